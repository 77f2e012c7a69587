#!/usr/bin/env python3
"""tools/adversarial_pairs.py FIRST COUNT [--procs P] [--out FILE]: the constructed parity cases of tests/adversarial_cases.py
(samples within ulps of knots, of the clamp at 255 and of 0, of the frame borders; cells at the 300-pixel threshold and
cells left with a handful of samples; steep edges under all of that) through both math modes and the four workgroup
shapes, against the oracle -- tools/parity_sweep_mp.py with --generator adversarial --shapes 0,256,512,1024.
profiles/r05_adversarial.txt is its output."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
args = sys.argv[1:]
if not any(a.startswith("--shapes") for a in args):
    args += ["--shapes", "0,256,512,1024"]
sys.exit(subprocess.call([sys.executable, os.path.join(here, "parity_sweep_mp.py"), *args, "--generator", "adversarial"]))
