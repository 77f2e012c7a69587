#!/bin/bash
mkdir -p gpurun_out/r04c
ROUNDS=2 python tools/flash_ab.py "$@" > gpurun_out/r04c/flash_ab_$$.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04c/flash_ab_$$.txt
