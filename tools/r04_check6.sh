#!/bin/bash
mkdir -p gpurun_out/r04h
DIAG_ORACLE=1 python tools/diag_forms.py 6 512 > gpurun_out/r04h/diag_forms.txt 2>&1; grep "flash\|oracle" gpurun_out/r04h/diag_forms.txt | head -12
python -m pytest tests -m gpu -q > gpurun_out/r04h/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r04h/pytest.log
ROUNDS=1 python tools/flash_ab.py exp/libnid_norepair.so default > gpurun_out/r04h/flash_ab.txt 2>&1; cat gpurun_out/r04h/flash_ab.txt
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04h/trace -- python3 $R/tools/trace_run.py > /dev/null 2>&1; echo "trace rc=$?"
cd $R; f=$(find gpurun_out/r04h/trace -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-220
