#!/bin/bash
# ARCHIVED: chunked driver of the single-process random sweep (rounds 2-3; superseded by tools/parity_sweep_mp.py); output recorded in / cited by: HISTORY.md (there as tools/parity_sweeps.sh). Not part of the test or measurement flow.
# tools/parity_sweeps.sh FIRST CHUNKS [CHUNK]: tools/random_parity_sweep.py over CHUNKS chunks of CHUNK seeds (default 200)
# at the throughput shape, the same seeds at 512 threads (latency form), then tools/flash_pose_sweep.py; one summary
# line per chunk (a GPU box kills a silent command after seven minutes).
first=$1; chunks=$2; chunk=${3:-200}
mkdir -p gpurun_out/sweeps
for ((c = 0; c < chunks; c++)); do
  f=$((first + c * chunk))
  timeout -k 10 400 python tools/archive/random_parity_sweep.py $f $chunk 2>&1 | tail -4 | tee -a gpurun_out/sweeps/random_128.txt
  timeout -k 10 400 python tools/archive/random_parity_sweep.py $f $chunk 512 2>&1 | tail -4 | tee -a gpurun_out/sweeps/random_512.txt
done
timeout -k 10 400 python tools/flash_pose_sweep.py 24 2>&1 | tail -8 | tee -a gpurun_out/sweeps/flash.txt
