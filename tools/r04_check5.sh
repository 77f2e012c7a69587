#!/bin/bash
mkdir -p gpurun_out/r04e
ROUNDS=2 python tools/flash_ab.py exp/libnid_norepair.so default > gpurun_out/r04e/flash_ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r04e/flash_ab.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r04e/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r04e/pytest.log
python tools/pair_setup.py A 8 > gpurun_out/r04e/pair_setup_A.txt 2>&1; echo "pair rc=$?"; cat gpurun_out/r04e/pair_setup_A.txt
python tools/short_seq_sweep.py A 8 > gpurun_out/r04e/short_seq_A.txt 2>&1; echo "sweep rc=$?"; grep "n  10\|n  20\|n  16\|n  32\|config" gpurun_out/r04e/short_seq_A.txt
