#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r04_parity_sweeps.txt, HISTORY.md (there as tools/diag_dump_cell.py). Not part of the test or measurement flow.
"""tools/diag_dump_cell.py SEED POSE OUT.npz: the FAST diagnostic kernel's per-pixel dumps (cost phase and Jacobian phase)
and per-cell results of one random parity case, saved for offline analysis (tools/diag_fast_terms.py, no GPU)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
import test_parity_gpu as T
seed, k, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pair, nb, poses = T._random_case(synth, 1000 + seed)
pose = poses[k]
save = {}
for math in ("fast", "strict"):
    ctx = capi.from_pair(pair, nb, math=T._mode(capi, math)); ctx.compute_href(pair.pose_init)
    plain = ctx.evaluate(pose, True)
    ctx.enable_pixel_dump(1); g1 = ctx.evaluate(pose, True); c = ctx.pixel_dump()
    ctx.enable_pixel_dump(2); g2 = ctx.evaluate(pose, True); j = ctx.pixel_dump()
    for n, v in c.items(): save[f"{math}_cost_{n}"] = v
    for n, v in j.items(): save[f"{math}_jac_{n}"] = v
    for n, v in zip(("Hc", "Hj", "err", "J"), plain): save[f"{math}_{n}"] = v
    save[f"{math}_dbgJ"] = g2[3]
    ctx.close()
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
np.savez_compressed(out, **save)
print("saved", out)
