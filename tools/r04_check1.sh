#!/bin/bash
# round 4, first GPU check: the suite, then the kernels the repair pass touched (throughput, flash, latency shapes)
mkdir -p gpurun_out/r04a
python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r04a/pytest.log
python tools/flash_rate.py > gpurun_out/r04a/flash_rate.txt 2>&1; echo "flash rc=$?"; cat gpurun_out/r04a/flash_rate.txt
python tools/direct_latency.py A 8 > gpurun_out/r04a/direct_latency_A.txt 2>&1; echo "direct rc=$?"; cat gpurun_out/r04a/direct_latency_A.txt
python tools/batch_sweep.py > gpurun_out/r04a/batch_sweep.txt 2>&1; echo "batch rc=$?"; cat gpurun_out/r04a/batch_sweep.txt
