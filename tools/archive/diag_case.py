#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r04_parity_sweeps.txt (there as tools/diag_case.py). Not part of the test or measurement flow.
"""tools/diag_case.py SEED [SHAPE...]: one random parity case (tests/test_parity_gpu.py::_random_case), per math mode and
workgroup shape: which cells miss the Jacobian bound, by how much, and how many cells ran the repair pass."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as oracle
import test_parity_gpu as T
seed = int(sys.argv[1])
shapes = [int(x) for x in sys.argv[2:]] or [0, 512]
pair, nb, poses = T._random_case(synth, 1000 + seed)
o = oracle.from_pair(pair, nb)
cnt_o, href_o = o.compute_href(pair.pose_init)
act = cnt_o >= 300
print(f"seed {seed}: {pair.rows}x{pair.cols}, {pair.cell}x{pair.cell} cells, nb {nb}, active {int(act.sum())}")
for pi, pose in enumerate(poses):
    ref = o.evaluate(pose, True)
    for math in T.MODES:
        for shape in shapes:
            ctx = capi.from_pair(pair, nb, math=T._mode(capi, math))
            if shape:
                ctx.set_launch_shape(shape, shape)
            ctx.compute_href(pair.pose_init)
            ctx.repair_count(reset=True)
            got = ctx.evaluate(pose, True)
            rep = ctx.repair_count()
            got2 = ctx.evaluate(pose, True)
            fin = np.isfinite(ref[3]).all(axis=1) & np.isfinite(got[3]).all(axis=1) & act
            line = f"  pose {pi} {math:6s} shape {shape:4d}: repairs {rep}"
            if fin.any():
                ex, percell = T._jac_excess(got[3], ref[3], fin)
                bad = np.where(fin)[0][ex > 1.0]
                line += f"; worst {ex.max() * T.RTOL_J:.3e}; cells over the bound {list(bad[:8])}"
                for c in bad[:3]:
                    line += f"\n      cell {c}: J {got[3][c]}\n             ref {ref[3][c]}  Hc/Hj/err diff {abs(got[0][c]-ref[0][c]):.1e} {abs(got[1][c]-ref[1][c]):.1e} {abs(got[2][c]-ref[2][c]):.1e}"
            line += f"; run-to-run same {all(np.array_equal(T._bits(a), T._bits(b)) for a, b in zip(got, got2))}"
            print(line)
            ctx.close()
