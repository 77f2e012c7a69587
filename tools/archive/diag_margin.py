#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r03_parity_sweeps.txt, HISTORY.md, profiles/README.md (there as tools/diag_margin.py). Not part of the test or measurement flow.
"""tools/diag_margin.py SEED [SEED...]: for sweep cases close to (or over) the 1e-9 Jacobian rule, where does the
difference come from?  Per seed, pose and math mode: the worst cell's |dJ| against the oracle in units of the plain
allowance (1e-9 of the cell's own scale + f64 roundoff at the frame's scale), the same figure for the ORACLE'S TWIN
against the oracle (two valid roundings of the reference's own expressions: its arithmetic noise, tests/test_parity_gpu.py
_reference_noise), and for FAST against STRICT (what the FAST mode's own arithmetic adds).  GPU needed."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
import test_parity_gpu as T

for seed in [int(x) for x in sys.argv[1:]]:
    pair, nb, poses = T._random_case(synth, 1000 + seed)
    o = O.from_pair(pair, nb)
    cnt, _ = o.compute_href(pair.pose_init)
    act = cnt >= 300
    tw = o.twin()
    tw.compute_href(pair.pose_init)
    ctxs = {}
    for math in T.MODES:
        ctxs[math] = capi.from_pair(pair, nb, math=T._mode(capi, math))
        ctxs[math].compute_href(pair.pose_init)
    print(f"seed {seed}: {pair.rows}x{pair.cols}, {nb} bins, grey levels ref/target "
          f"{len(np.unique(pair.im0))}/{len(np.unique(pair.im1))}, active cells {int(act.sum())}")
    for k, pose in enumerate(poses):
        ref = o.evaluate(pose, True)
        twin = tw.evaluate(pose, True)
        got = {m: c.evaluate(pose, True) for m, c in ctxs.items()}
        fin = np.isfinite(ref[3]).all(axis=1) & act
        for m in got:
            fin &= np.isfinite(got[m][3]).all(axis=1)
        fin &= np.isfinite(twin[3]).all(axis=1)
        if not fin.any():
            continue
        line = [f"  pose {k}:"]
        for name, a in [(m, got[m][3]) for m in got] + [("twin", twin[3])]:
            ex = T._jac_excess(a, ref[3], fin)[0]
            c = int(np.argmax(ex))
            line.append(f"{name} {ex[c]:.3f} (cell {np.flatnonzero(fin)[c]})")
        ms = list(got)
        if len(ms) == 2:
            ex = T._jac_excess(got[ms[0]][3], got[ms[1]][3], fin)[0]
            line.append(f"{ms[0]}-vs-{ms[1]} {ex.max():.3f}")
        print(" ".join(line) + "   [units of the plain allowance; 1.0 = the 1e-9 rule]")
