#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r04_parity_sweeps.txt (there as tools/diag_jac_pixels.py). Not part of the test or measurement flow.
"""tools/diag_jac_pixels.py SEED POSE CELL: which pixels contribute to the Jacobian on the GPU (FAST, diagnostic kernel,
Jacobian-phase dump) and in the oracle."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
import test_parity_gpu as T
seed, k, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pair, nb, poses = T._random_case(synth, 1000 + seed)
pose = poses[k]
o = O.from_pair(pair, nb); o.compute_href(pair.pose_init); ref = o.evaluate(pose, True)
d, j = o.dump_pixels(), o.dump_jac()
G, rb, cb = pair.cell, pair.rows // pair.cell, pair.cols // pair.cell
rr, cc = np.divmod(np.arange(pair.rows * pair.cols), pair.cols)
cell = np.where((rr < G * rb) & (cc < G * cb), (rr // rb) * G + cc // cb, -1)
ids = np.where(cell == c)[0]
ctx = capi.from_pair(pair, nb); ctx.compute_href(pair.pose_init)
ctx.enable_pixel_dump(2)
got = ctx.evaluate(pose, True)
g = ctx.pixel_dump()
print("dbg kernel J", got[3][c], "ref", ref[3][c])
gpu_on = g["jc"][ids] >= 0
orc_on = j["jc"][ids] >= 0
print("pixels in cell", len(ids), "gpu jac", gpu_on.sum(), "oracle jac", orc_on.sum(), "cost (oracle)", (d["jc"][ids] >= 0).sum())
extra = ids[gpu_on & ~orc_on]; missing = ids[~gpu_on & orc_on]
print("gpu only:", len(extra), [(int(rr[i]), int(cc[i]), float(d["u"][i]), float(d["v"][i])) for i in extra[:10]])
print("oracle only:", len(missing), [(int(rr[i]), int(cc[i]), float(d["u"][i]), float(d["v"][i])) for i in missing[:10]])
both = ids[gpu_on & orc_on]
dg = np.abs(g["u"][both] - j["gx"][both]); print("max |gx diff|", dg.max() if len(both) else None, "max |gy diff|", np.abs(g["v"][both] - j["gy"][both]).max() if len(both) else None,
      "max |pc diff|", np.abs(g["ic"][both] - j["pc"][both]).max() if len(both) else None, "jc mismatches", int((g["jc"][both] != j["jc"][both]).sum()))
mm = both[g["jc"][both] != j["jc"][both]]
for i in mm[:5]:
    print("jc mismatch at", int(rr[i]), int(cc[i]), "gpu pc", repr(float(g["ic"][i])), "jc", int(g["jc"][i]), "dw", g["wc"][i], "| oracle pc", repr(float(j["pc"][i])), "jc", int(j["jc"][i]), "dw", j["dw"][i],
          "ic", repr(float(d["ic"][i])), "wr", d["wr"][i], "jr", int(d["jr"][i]), "gx gy", j["gx"][i], j["gy"][i])
big = both[np.argsort(-np.abs(g["wc"][both] - j["dw"][both]).max(axis=1))[:3]]
for i in big:
    print("largest dw diff at", int(rr[i]), int(cc[i]), "gpu", g["wc"][i], "oracle", j["dw"][i], "pc", repr(float(j["pc"][i])))
