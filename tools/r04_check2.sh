#!/bin/bash
# round 4, second GPU check: same-box A/B of the kernels (round 3's, the repair build, the same without the repair code),
# the short-sequence sweep, the multi-rank bench tests, the driver-style bench line
mkdir -p gpurun_out/r04b
ROUNDS=2 python tools/flash_ab.py exp/libnid_r03.so default exp/libnid_norepair.so > gpurun_out/r04b/flash_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04b/flash_ab.txt
python tools/short_seq_sweep.py A 8 > gpurun_out/r04b/short_seq_A.txt 2>&1; echo "sweep rc=$?"; cat gpurun_out/r04b/short_seq_A.txt
python -m pytest tests/test_host_gpu.py -m gpu -x -q -k "bench_multi_rank or full_batches or launch_chain" > gpurun_out/r04b/pytest_bench.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r04b/pytest_bench.log
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "full_batches or launch_chain or batched" > gpurun_out/r04b/pytest_batches.log 2>&1; echo "pytest2 rc=$?"; tail -3 gpurun_out/r04b/pytest_batches.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04b/bench_driver.json 2> gpurun_out/r04b/bench_driver.err; echo "bench rc=$?"; python tools/bench_summary.py gpurun_out/r04b/bench_driver.json 2>/dev/null | head -30
