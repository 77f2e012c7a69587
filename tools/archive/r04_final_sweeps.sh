#!/bin/bash
# ARCHIVED: round 4's sweep / check driver script (its output is under profiles/r04_*); output recorded in / cited by: profiles/r04_parity_sweeps.txt, profiles/README.md (there as tools/r04_final_sweeps.sh). Not part of the test or measurement flow.
# tools/r04_final_sweeps.sh PART: the parity sweeps of profiles/r04_parity_sweeps.txt on the FINAL kernel.
# PART 1: round 3's ranges (13 000 cases) + this round's first two ranges (14 600); PART 2: 20 600 fresh cases + the flash sweep.
out=gpurun_out/r04_sweeps
mkdir -p $out
run() { timeout -k 10 $(( $2 / 30 + 150 )) python tools/parity_sweep_mp.py $1 $2 --procs 5 --shapes $3 --out $out/sweeps_part$PART.txt 2>&1 | grep -v "^\[w[0-9]*\] [0-9]*/" | grep -v Warning | grep -v "d = np" | tail -4 | cut -c1-400; }
PART=$1
if [ "$PART" = 1 ]; then
  run 300000 12000 0,512
  run 320000 1000 256,1024
  run 500000 14000 0,512
  run 517000 600 256,1024
else
  run 520000 20000 0,512
  run 540000 600 256,1024
  timeout -k 10 300 python tools/flash_pose_sweep.py 24 2>&1 | tail -6 | tee $out/flash.txt
fi
