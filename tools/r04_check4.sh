#!/bin/bash
mkdir -p gpurun_out/r04d
ROUNDS=2 python tools/flash_ab.py exp/libnid_norepair.so default > gpurun_out/r04d/flash_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04d/flash_ab.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r04d/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r04d/pytest.log
python tools/direct_latency.py A 8 > gpurun_out/r04d/direct_latency_A.txt 2>&1; echo "direct rc=$?"; cat gpurun_out/r04d/direct_latency_A.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r04d/bench_driver.json 2> gpurun_out/r04d/bench_driver.err; echo "bench rc=$?"; python tools/bench_summary.py gpurun_out/r04d/bench_driver.json 2>/dev/null | head -30
