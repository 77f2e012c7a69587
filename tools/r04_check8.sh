#!/bin/bash
mkdir -p gpurun_out/r04j
ROUNDS=1 python tools/flash_ab.py "$@" > gpurun_out/r04j/ab_$$.txt 2>&1; cat gpurun_out/r04j/ab_$$.txt
