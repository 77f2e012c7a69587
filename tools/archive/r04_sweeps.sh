#!/bin/bash
# ARCHIVED: round 4's sweep / check driver script (its output is under profiles/r04_*); output recorded in / cited by: profiles/r04_parity_sweeps.txt (there as tools/r04_sweeps.sh). Not part of the test or measurement flow.
# round 4 parity sweeps on the final kernel: fresh random cases (throughput shape and the 512-thread latency form, both math
# modes), 256 / 1024 threads on a smaller set, then the flash sweep.  usage: tools/r04_sweeps.sh FIRST COUNT [OUTDIR]
out=${3:-gpurun_out/r04s}
mkdir -p $out
t=$(( $2 / 30 + 150 ))
timeout -k 10 $t python tools/parity_sweep_mp.py $1 $2 --procs 5 --shapes 0,512 --out $out/sweeps.txt 2>&1 | grep -v "^\[w[0-9]*\] [0-9]*/" | grep -v Warning | grep -v "d = np" | tail -8
timeout -k 10 200 python tools/parity_sweep_mp.py $(($1 + $2)) 600 --procs 5 --shapes 256,1024 --out $out/sweeps.txt 2>&1 | grep -v "^\[w[0-9]*\] [0-9]*/" | grep -v Warning | grep -v "d = np" | tail -5
timeout -k 10 300 python tools/flash_pose_sweep.py 24 2>&1 | tail -6 | tee -a $out/flash.txt
