#!/bin/bash
# ARCHIVED: round 4's sweep / check driver script (its output is under profiles/r04_*); output recorded in / cited by: profiles/r04_parity_sweeps.txt (there as tools/r04_extra_sweeps.sh). Not part of the test or measurement flow.
# tools/r04_extra_sweeps.sh FIRST COUNT [SHAPES]: more fresh cases on the final kernel (appended to profiles/r04_parity_sweeps.txt).
# (the workers' progress lines go to a file under gpurun_out/ as they come: a run that is silent for 7 minutes is taken for hung)
out=gpurun_out/r04_sweeps
shapes=${3:-0,512}
mkdir -p $out
timeout -k 10 $(( $2 / 30 + 150 )) python tools/parity_sweep_mp.py $1 $2 --procs 5 --shapes $shapes --out $out/sweeps_extra.txt 2>&1 | tee -a $out/progress.log | grep -v "^\[w[0-9]*\] [0-9]*/" | grep -v Warning | grep -v "d = np" | tail -4 | cut -c1-600
