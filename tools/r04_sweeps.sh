#!/bin/bash
# round 4 parity sweeps on the final kernel: fresh random cases (throughput shape and the 512-thread latency form, both math
# modes), 256 / 1024 threads on a smaller set, then the flash sweep
mkdir -p gpurun_out/r04s
timeout -k 10 520 python tools/parity_sweep_mp.py $1 $2 --procs 5 --shapes 0,512 --out gpurun_out/r04s/sweeps.txt 2>&1 | grep -v "^\[w" | tail -3
timeout -k 10 200 python tools/parity_sweep_mp.py $(($1 + $2)) 600 --procs 5 --shapes 256,1024 --out gpurun_out/r04s/sweeps.txt 2>&1 | grep -v "^\[w" | tail -3
timeout -k 10 300 python tools/flash_pose_sweep.py 24 2>&1 | tail -6 | tee -a gpurun_out/r04s/flash.txt
