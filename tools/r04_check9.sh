#!/bin/bash
mkdir -p gpurun_out/r04t
for s in 502812 503912 500953 500630; do python tools/diag_case.py $s 0 512 2>&1 | grep "worst" | awk '{print $0}' | sort -t' ' -k9 -g | tail -1; done
python -m pytest tests -m gpu -q -x > gpurun_out/r04t/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r04t/pytest.log
ROUNDS=2 python tools/flash_ab.py exp/libnid_norepair.so default > gpurun_out/r04t/ab.txt 2>&1; cat gpurun_out/r04t/ab.txt
