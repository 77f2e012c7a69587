#!/bin/bash
mkdir -p gpurun_out/r04i
python -m pytest tests -m gpu -q -x > gpurun_out/r04i/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/r04i/pytest.log
R=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04i/trace_cur -- python3 $R/tools/trace_run.py plain > /dev/null 2>&1; echo "trace rc=$?"
export NID_HIP_LIB=$R/exp/libnid_norepair.so
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04i/trace_norep -- python3 $R/tools/trace_run.py plain > /dev/null 2>&1; echo "trace2 rc=$?"
unset NID_HIP_LIB
cd $R
