#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; kept because profiles/ and HISTORY.md cite its output (as tools/diag_cell_j.py). Not part of the test or measurement flow.
"""tools/diag_cell_j.py SEED POSE_INDEX CELL: recompute one cell's Jacobian in long double from the ORACLE's per-pixel dumps
(weights, gradients, B-spline derivatives) and compare with what the oracle and the HIP path return -- to tell a
summation-order / cancellation effect in the reference's own f64 evaluation from an error of the kernel."""
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
LD = np.longdouble
o = O.from_pair(pair, nb)
cnt_o, href_o = o.compute_href(pair.pose_init)
ref = o.evaluate(pose, True)
d, j = o.dump_pixels(), o.dump_jac()
G, rb, cb = pair.cell, pair.rows // pair.cell, pair.cols // pair.cell
rr, cc = np.divmod(np.arange(pair.rows * pair.cols), pair.cols)
cell = np.where((rr < G * rb) & (cc < G * cb), (rr // rb) * G + cc // cb, -1)
ids = np.where(cell == c)[0]
Nc = LD(cnt_o[c]); S = nb - 3
hc = np.zeros(nb, dtype=LD); hj = np.zeros((nb, nb), dtype=LD)
for i in ids:
    if d["jc"][i] < 0: continue
    jc, jr = d["jc"][i], d["jr"][i]
    for kk in range(4):
        hc[jc + kk] += LD(d["wc"][i, kk])
        for m in range(4):
            hj[jr + m, jc + kk] += LD(d["wr"][i, m]) * LD(d["wc"][i, kk])
pc_, pj_ = hc / Nc, hj / Nc
sig = LD(1e-30)
def ent_w(p):
    w = np.zeros_like(p); e = LD(0)
    it = np.nditer(p, flags=["multi_index"])
    for x in it:
        x = LD(x)
        if not (x < sig):
            l = np.log2(x); w[it.multi_index] = -(1 + l); e -= x * l
    return e, w
Hc, Wc = ent_w(pc_); Hj, Wj = ent_w(pj_)
print(f"seed {seed} pose {k} cell {c}: nb {nb}, N_c {int(Nc)}, Hc {float(Hc):.15f} (oracle {ref[0][c]:.15f})  Hj {float(Hj):.15f} (oracle {ref[1][c]:.15f})")
small = [(float(pj_[a, b]), a, b) for a in range(nb) for b in range(nb) if 0 < pj_[a, b] < 1e-6]
print("joint bins with 0 < p < 1e-6:", [(f"{p:.3e}", a, b, f"W {float(Wj[a,b]):.1f}") for p, a, b in sorted(small)[:12]])
smallc = [(float(pc_[b]), b) for b in range(nb) if 0 < pc_[b] < 1e-6]
print("target bins with 0 < p < 1e-6:", [(f"{p:.3e}", b) for p, b in smallc])
# camera-frame points
M = synth.pose7_to_matrix(pose)
z0 = pair.depth_m.reshape(-1)
x0 = z0 * (cc - pair.cx) / pair.fx; y0 = z0 * (rr - pair.cy) / pair.fy
Xw = (pair.T_wc0 @ np.stack([x0, y0, z0, np.ones_like(z0)]))[:3]
Xc = M[:3, :3] @ Xw + M[:3, 3:4]
href = LD(href_o[c])
acc = np.zeros(6, dtype=LD)
terms = []
for i in ids:
    if j["jc"][i] < 0: continue
    jc, jr = j["jc"][i], d["jr"][i]
    dw = j["dw"][i].astype(LD)
    t = sum(Wc[jc + m] * dw[m] for m in range(4))
    s = sum(LD(d["wr"][i, kk]) * sum(Wj[jr + kk, jc + m] * dw[m] for m in range(4)) for kk in range(4))
    cf = s * (Hc + href) - t * Hj
    x, y, z = (LD(v) for v in Xc[:, i]); a, b, iz = x / z, y / z, 1 / z
    Ju = LD(pair.fx) * np.array([-a * b, 1 + a * a, -b, iz, 0, -a * iz], dtype=LD)
    Jv = LD(pair.fy) * np.array([-(1 + b * b), a * b, a, 0, iz, -b * iz], dtype=LD)
    dI = LD(j["gx"][i]) * Ju + LD(j["gy"][i]) * Jv
    acc += cf * dI
    terms.append(float(np.abs(cf * dI).max()))
J = acc * (LD(S) / 255) / Nc / (Hj * Hj)
ctx = capi.from_pair(pair, nb, math=capi.MATH_STRICT)
ctx.compute_href(pair.pose_init)
got = ctx.evaluate(pose, True)
print("long double :", np.array(J, dtype=np.float64))
print("oracle      :", ref[3][c])
print("HIP (STRICT):", got[3][c])
print("|oracle - ld| max %.3e   |HIP - ld| max %.3e   largest single-pixel term %.3e x %d pixels" % (
    np.abs(ref[3][c] - np.array(J, dtype=np.float64)).max(), np.abs(got[3][c] - np.array(J, dtype=np.float64)).max(),
    max(terms) * float(S / 255 / Nc / (Hj * Hj)), len(terms)))
