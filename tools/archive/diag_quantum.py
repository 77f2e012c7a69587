#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r03_parity_sweeps.txt, HISTORY.md, tests/test_parity_gpu.py (there as tools/diag_quantum.py). Not part of the test or measurement flow.
"""tools/diag_quantum.py SEED POSE_INDEX CELL [QBITS...]: what the fixed-point quantum of the histogram adds does to one
cell's Jacobian -- no GPU needed.  The cell's Jacobian is recomputed in long double from the ORACLE's per-pixel dumps
(i) with exact histograms and (ii) with every coarse addend rounded to a multiple of 2^-q the way k_eval2's hist_add
does (samples next to a knot -- smaller outer weight below 2^-28, or tiny reference weights -- keep their small
weights exactly: the fine levels), for each q given (default 45 52 60)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
import test_parity_gpu as T
seed, k, c = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
qs = [int(x) for x in sys.argv[4:]] or [45, 52, 60]
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
TINY, FINE = 2.0 ** -35, 2.0 ** -8
M = synth.pose7_to_matrix(pose)
z0 = pair.depth_m.reshape(-1)
x0 = z0 * (cc - pair.cx) / pair.fx; y0 = z0 * (rr - pair.cy) / pair.fy
Xw = (pair.T_wc0 @ np.stack([x0, y0, z0, np.ones_like(z0)]))[:3]
Xc = M[:3, :3] @ Xw + M[:3, 3:4]
href = LD(href_o[c])


def quant(x, q):
    if q is None:
        return LD(x)
    return LD(np.rint(np.float64(x) * 2.0 ** q)) / LD(2.0) ** q   # x * 2^q < 2^53: exact scaling, RN-even


REPAIR = os.environ.get("DIAG_REPAIR")  # model of the round-4 repair pass (hist_add REPAIR in csrc/nid_kernels.hip.h)
LIN_FLAG, REPAIR_MASS = 2.0 ** -16, 2.0 ** -12


def fine_quant(x):
    """a fine-level add: level = floor((-8 - e) / 24) capped at 4, scale 2^(59 + 24 level), round to nearest"""
    if x <= 0: return LD(0)
    e = int(np.frexp(np.float64(x))[1])
    lv = min(max(-8 - e, 0), 119) // 24
    sc = 59 + 24 * lv
    return LD(np.rint(np.float64(x) * 2.0 ** sc)) / LD(2.0) ** sc


def jac(q):
    hc = np.zeros(nb, dtype=LD); hj = np.zeros((nb, nb), dtype=LD)
    coarse_j = np.zeros((nb, nb), dtype=bool); coarse_c = np.zeros(nb, dtype=bool)
    flag_j = np.zeros((nb, nb), dtype=bool); flag_c = np.zeros(nb, dtype=bool)
    def tiny_of(i):
        wc, wr = d["wc"][i], d["wr"][i]
        wrp = wr[wr > 0]
        return min(wc[0], wc[3]) < TINY or (wrp.size and min(wr[0], wr[3]) < TINY and min(wr[0], wr[3]) != 0.0)
    for i in ids:
        if d["jc"][i] < 0: continue
        jc, jr = d["jc"][i], d["jr"][i]
        wc, wr = d["wc"][i], d["wr"][i]
        tiny = tiny_of(i)
        for kk in range(4):
            fine_c = tiny and wc[kk] < FINE
            lin = (jc == 0 and kk == 1) or (jc == S - 1 and kk == 2)
            if fine_c and lin and 0 < wc[kk] < LIN_FLAG:
                flag_c[jc + kk] = True
                for m in range(4):
                    if wr[m] * wc[kk] > 0: flag_j[jr + m, jc + kk] = True
            hc[jc + kk] += LD(wc[kk]) if fine_c else quant(wc[kk], q)
            if not fine_c: coarse_c[jc + kk] = True
            for m in range(4):
                pr = np.float64(wr[m]) * np.float64(wc[kk])
                fine = tiny and (wc[kk] < FINE or wr[m] < FINE)
                hj[jr + m, jc + kk] += LD(pr) if (fine or q is None) else quant(pr, q)
                if not fine and pr != 0: coarse_j[jr + m, jc + kk] = True
    if REPAIR and q is not None:
        rep_j = flag_j & coarse_c[None, :] & (hj < REPAIR_MASS); rep_c = flag_c & coarse_c & (hc < REPAIR_MASS)
        if rep_j.any() or rep_c.any():
            print(f"    repair pass: joint bins {list(zip(*np.where(rep_j)))} marginal {list(np.where(rep_c)[0])}")
            hj[rep_j] = 0; hc[rep_c] = 0
            for i in ids:
                if d["jc"][i] < 0: continue
                jc, jr = d["jc"][i], d["jr"][i]
                wc, wr = d["wc"][i], d["wr"][i]
                tiny = tiny_of(i)
                for kk in range(4):
                    fine_c = tiny and wc[kk] < FINE
                    if rep_c[jc + kk]: hc[jc + kk] += LD(wc[kk]) if fine_c else fine_quant(wc[kk])
                    for m in range(4):
                        if not rep_j[jr + m, jc + kk]: continue
                        pr = np.float64(wr[m]) * np.float64(wc[kk])
                        fine = tiny and (wc[kk] < FINE or wr[m] < FINE)
                        hj[jr + m, jc + kk] += LD(pr) if fine else fine_quant(pr)
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
    acc = np.zeros(6, dtype=LD)
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
        acc += cf * (LD(j["gx"][i]) * Ju + LD(j["gy"][i]) * Jv)
    return np.array(acc * (LD(S) / 255) / Nc / (Hj * Hj), dtype=np.float64)


J0 = jac(None)
print(f"seed {seed} pose {k} cell {c}: nb {nb} N_c {int(Nc)}  max|J| {np.abs(J0).max():.3e}   |oracle - exact| {np.abs(ref[3][c] - J0).max():.3e}")
for q in qs:
    Jq = jac(q)
    print(f"  quantum 2^-{q}: |J_q - exact| max {np.abs(Jq - J0).max():.3e}  = {np.abs(Jq - J0).max() / np.abs(J0).max():.3e} of the cell's scale")
if os.environ.get("DIAG_BINS"):
    # per-pixel weights of the Jacobian samples
    for i in ids:
        if j["jc"][i] < 0: continue
        print("  jac sample: jr", d["jr"][i], "jc", j["jc"][i], "wr", d["wr"][i], "wc", d["wc"][i], "dw", j["dw"][i])
if os.environ.get("DIAG_ATTR"):
    # which bins carry the difference: masses exact vs quantised, and each bin's share of the Jacobian change
    def hists(q):
        hc = np.zeros(nb, dtype=LD); hj = np.zeros((nb, nb), dtype=LD)
        for i in ids:
            if d["jc"][i] < 0: continue
            jc, jr = d["jc"][i], d["jr"][i]
            wc, wr = d["wc"][i], d["wr"][i]
            wrp = wr[wr > 0]
            tiny = min(wc[0], wc[3]) < TINY or (wrp.size and min(wr[0], wr[3]) < TINY and min(wr[0], wr[3]) != 0.0)
            for kk in range(4):
                fine_c = tiny and wc[kk] < FINE
                hc[jc + kk] += LD(wc[kk]) if fine_c else quant(wc[kk], q)
                for m in range(4):
                    pr = np.float64(wr[m]) * np.float64(wc[kk])
                    fine = tiny and (wc[kk] < FINE or wr[m] < FINE)
                    hj[jr + m, jc + kk] += LD(pr) if (fine or q is None) else quant(pr, q)
        return hc, hj
    hc0, hj0 = hists(None); hcq, hjq = hists(qs[0])
    rel = np.where(hj0 > 0, np.abs(hjq - hj0) / np.where(hj0 > 0, hj0, 1), 0)
    order = np.dstack(np.unravel_index(np.argsort(-rel.ravel()), rel.shape))[0][:8]
    for r_, c_ in order:
        print(f"  joint bin ({r_},{c_}): mass {float(hj0[r_, c_]):.4e}  relative change {float(rel[r_, c_]):.3e}")
    relc = np.where(hc0 > 0, np.abs(hcq - hc0) / np.where(hc0 > 0, hc0, 1), 0)
    for b in np.argsort(-relc)[:4]:
        print(f"  marginal bin {b}: mass {float(hc0[b]):.4e}  relative change {float(relc[b]):.3e}")
