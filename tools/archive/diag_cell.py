#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; kept because profiles/ and HISTORY.md cite its output (as tools/diag_cell.py). Not part of the test or measurement flow.
"""tools/diag_cell.py SEED POSE LIB_A LIB_B: per-cell outputs of two builds of the library on one sweep case, next to the oracle."""
import importlib, os, sys, subprocess, json
import numpy as np
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 4:   # parent: one child per library (the library is chosen at import time)
    for lib in sys.argv[3:]:
        env = dict(os.environ, NID_HIP_LIB=lib)
        subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], lib], env=env)
    sys.exit(0)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as oracle
import test_parity_gpu as T
seed, k = int(sys.argv[1]), int(sys.argv[2])
pair, nb, poses = T._random_case(synth, 1000 + seed)
o = oracle.from_pair(pair, nb)
cnt, _ = o.compute_href(pair.pose_init)
ref = o.evaluate(poses[k], True)
ctx = capi.from_pair(pair, nb)
ctx.compute_href(pair.pose_init)
got = ctx.evaluate(poses[k], True)
np.set_printoptions(precision=3, linewidth=200)
print("==", sys.argv[3])
for c in range(len(cnt)):
    if cnt[c] < 300: continue
    dj = np.abs(got[3][c] - ref[3][c]).max()
    print(f"cell {c:3d} n {cnt[c]:5d} dHc {got[0][c]-ref[0][c]:+.2e} dHj {got[1][c]-ref[1][c]:+.2e} derr {got[2][c]-ref[2][c]:+.2e} max|J_o| {np.abs(ref[3][c]).max():.3e} max|dJ| {dj:.2e}")
