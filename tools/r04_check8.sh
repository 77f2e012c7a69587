#!/bin/bash
mkdir -p gpurun_out/r04j
ROUNDS=2 python tools/flash_ab.py exp/libnid_norepair.so exp/libnid_st2k.so default > gpurun_out/r04j/ab1.txt 2>&1; cat gpurun_out/r04j/ab1.txt
