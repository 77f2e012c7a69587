#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r04_parity_sweeps.txt, HISTORY.md (there as tools/diag_fast_terms.py). Not part of the test or measurement flow.
"""tools/diag_fast_terms.py SEED POSE CELL DUMP.npz: which of FAST math's per-sample differences from the reference moves
one cell's Jacobian -- no GPU.  The cell's Jacobian is recomputed in long double from the ORACLE's per-pixel dumps with
one family of values at a time replaced by the FAST diagnostic kernel's (tools/diag_dump_cell.py): the histogram
weights, the Jacobian phase's derivative weights, its image gradient."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
import test_parity_gpu as T
seed, k, c, dump = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), np.load(sys.argv[4])
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
M = synth.pose7_to_matrix(pose)
z0 = pair.depth_m.reshape(-1)
x0 = z0 * (cc - pair.cx) / pair.fx; y0 = z0 * (rr - pair.cy) / pair.fy
Xw = (pair.T_wc0 @ np.stack([x0, y0, z0, np.ones_like(z0)]))[:3]
Xc = M[:3, :3] @ Xw + M[:3, 3:4]
href = LD(href_o[c])


def jac(wc_src, dw_src, g_src, only=None):
    """wc_src / dw_src / g_src: 'o' (oracle) or 'f' (FAST dump); only: restrict the FAST substitution to these pixel ids"""
    def pick(src, i, o_arr, f_arr):
        return f_arr[i] if (src == "f" and (only is None or i in only)) else o_arr[i]
    hc = np.zeros(nb, dtype=LD); hj = np.zeros((nb, nb), dtype=LD)
    for i in ids:
        if d["jc"][i] < 0: continue
        jc, jr = d["jc"][i], d["jr"][i]
        wc = pick(wc_src, i, d["wc"], dump["fast_cost_wc"]); wr = d["wr"][i]
        for kk in range(4):
            hc[jc + kk] += LD(wc[kk])
            for m in range(4):
                hj[jr + m, jc + kk] += LD(wr[m]) * LD(wc[kk])
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
        dw = pick(dw_src, i, j["dw"], dump["fast_jac_wc"]).astype(LD)
        gx = pick(g_src, i, j["gx"], dump["fast_jac_u"]); gy = pick(g_src, i, j["gy"], dump["fast_jac_v"])
        t = sum(Wc[jc + m] * dw[m] for m in range(4))
        s = sum(LD(d["wr"][i, kk]) * sum(Wj[jr + kk, jc + m] * dw[m] for m in range(4)) for kk in range(4))
        cf = s * (Hc + href) - t * Hj
        x, y, z = (LD(v) for v in Xc[:, i]); a, b, iz = x / z, y / z, 1 / z
        Ju = LD(pair.fx) * np.array([-a * b, 1 + a * a, -b, iz, 0, -a * iz], dtype=LD)
        Jv = LD(pair.fy) * np.array([-(1 + b * b), a * b, a, 0, iz, -b * iz], dtype=LD)
        acc += cf * (LD(gx) * Ju + LD(gy) * Jv)
    return np.array(acc * (LD(S) / 255) / Nc / (Hj * Hj), dtype=np.float64)


J0 = jac("o", "o", "o")
sc = np.abs(J0).max()
print(f"seed {seed} pose {k} cell {c}: |oracle - exact| {np.abs(ref[3][c] - J0).max() / sc:.2e}; FAST kernel - oracle {np.abs(dump['fast_J'][c] - ref[3][c]).max() / sc:.3e} "
      f"(diagnostic kernel {np.abs(dump['fast_dbgJ'][c] - ref[3][c]).max() / sc:.3e}) of the cell's scale")
same_jc = np.array_equal(dump["fast_cost_jc"][ids], d["jc"][ids]) and np.array_equal(dump["fast_jac_jc"][ids], j["jc"][ids])
print("   spans equal in both phases:", same_jc)
for name, a in (("histogram weights", ("f", "o", "o")), ("derivative weights", ("o", "f", "o")), ("image gradient", ("o", "o", "f")), ("all three", ("f", "f", "f"))):
    print(f"   FAST {name:18s}: |J - exact| {np.abs(jac(*a) - J0).max() / sc:.3e}")
if os.environ.get("DIAG_SAMPLES"):
    # the samples whose gradient substitution matters most
    on = [i for i in ids if j["jc"][i] >= 0]
    dg = np.array([max(abs(dump["fast_jac_u"][i] - j["gx"][i]), abs(dump["fast_jac_v"][i] - j["gy"][i])) for i in on])
    for w in np.argsort(-dg)[:6]:
        i = on[w]
        print(f"   sample ({rr[i]},{cc[i]}): gx {j['gx'][i]!r} FAST {dump['fast_jac_u'][i]!r} gy {j['gy'][i]!r} FAST {dump['fast_jac_v'][i]!r} u {d['u'][i]!r} v {d['v'][i]!r} "
              f"alone moves J by {np.abs(jac('o', 'o', 'f', only={i}) - J0).max() / sc:.3e}")
if os.environ.get("DIAG_WEIGHT_SAMPLES"):
    on = [i for i in ids if d["jc"][i] >= 0]
    cl = {i for i in on if d["ic"][i] == 254.999}
    print(f"   histogram weights of the {len(cl)} clamped samples only: {np.abs(jac('f', 'o', 'o', only=cl) - J0).max() / sc:.3e}; of the other {len(on) - len(cl)}: "
          f"{np.abs(jac('f', 'o', 'o', only=set(on) - cl) - J0).max() / sc:.3e}")
    rel = np.array([np.max(np.abs(dump["fast_cost_wc"][i] - d["wc"][i])) for i in on])
    for w in np.argsort(-rel)[:8]:
        i = on[w]
        print(f"   sample ({rr[i]},{cc[i]}): ic {d['ic'][i]!r} FAST {dump['fast_cost_ic'][i]!r} jc {d['jc'][i]} wc {d['wc'][i]} FAST - oracle {dump['fast_cost_wc'][i] - d['wc'][i]} "
              f"alone: {np.abs(jac('f', 'o', 'o', only={i}) - J0).max() / sc:.3e}")
