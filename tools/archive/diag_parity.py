#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; kept because profiles/ and HISTORY.md cite its output (as tools/diag_parity.py). Not part of the test or measurement flow.
"""tools/diag_parity.py [pair-spec ...]: HIP (FAST and STRICT) against the oracle on a pair, with the worst cells
and, for the worst cell, the per-pixel differences.  Diagnostic for the parity tests (GPU box)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O


def run(pair, nb, pose, label, xform="quat", href_pose=None, jacdump=False):
    o = O.from_pair(pair, nb, xform=xform)
    cnt_o, href_o = o.compute_href(pair.pose_init if href_pose is None else href_pose)
    ref = o.evaluate(pose, True)
    d = o.dump_pixels()
    act = cnt_o >= 300
    G, rb, cb = pair.cell, pair.rows // pair.cell, pair.cols // pair.cell
    rr, cc = np.divmod(np.arange(pair.rows * pair.cols), pair.cols)
    cell = (rr // rb) * G + cc // cb
    for mode, name in ((capi.MATH_FAST, "FAST"), (capi.MATH_STRICT, "STRICT")):
        ctx = capi.from_pair(pair, nb, math=mode, xform=capi.XFORM_QUAT if xform == "quat" else capi.XFORM_MATRIX)
        cnt, href = ctx.compute_href(pair.pose_init if href_pose is None else href_pose)
        ctx.enable_pixel_dump(True)
        got = ctx.evaluate(pose, True)
        g = ctx.pixel_dump()
        ctx.enable_pixel_dump(False)
        dH = [np.abs(got[k][act] - ref[k][act]) for k in range(3)]
        fin = np.isfinite(ref[3]).all(axis=1) & act
        pc = np.abs(ref[3][fin]).max(axis=1)
        sc = np.maximum(pc, 1e-6 * pc.max())
        dJ = (np.abs(got[3][fin] - ref[3][fin]) / sc[:, None]).max(axis=1)
        print(f"[{label} nb={nb} {name}] cnt equal {np.array_equal(cnt, cnt_o)}  max|dHc| {np.nanmax(dH[0]):.3e} |dHj| {np.nanmax(dH[1]):.3e} "
              f"|derr| {np.nanmax(dH[2]):.3e}  worst per-cell rel J {dJ.max():.3e}  cells over 1e-11: {(np.nan_to_num(dH[2]) > 1e-11).sum()}"
              f"  cells J over 1e-9: {(dJ > 1e-9).sum()}")
        idf = np.where(fin)[0]
        for c in idf[np.argsort(-dJ)[:4]]:
            k = np.where(idf == c)[0][0]
            print(f"    cell {c}: max|J_o| {np.abs(ref[3][c]).max():.3e} (global {pc.max():.3e})  max|dJ| {np.abs(got[3][c] - ref[3][c]).max():.3e}"
                  f"  rel-to-cell {dJ[k]:.2e}  Hc {ref[0][c]:.4f} Hj {ref[1][c]:.4f} err {ref[2][c]:.6f}  saturated px {int(((d['ic'] > 254.99) & (cell == c)).sum())}")
        ids = np.where(act)[0]
        worst = ids[np.nanargmax(np.maximum(dH[0], dH[1]))]
        m = (cell == worst) & ~np.isnan(d["u"])
        inb_o, inb_g = d["jc"][m] >= 0, g["jc"][m] >= 0
        both = inb_o & inb_g
        dic = np.abs(g["ic"][m][both] - d["ic"][m][both])
        print(f"    worst cell {worst}: in-frame oracle {inb_o.sum()} hip {inb_g.sum()} mismatched {np.sum(inb_o != inb_g)}; "
              f"max|dic| {dic.max() if dic.size else 0:.3e}; pixels with |dic| > 1e-9: {(dic > 1e-9).sum()}; "
              f"jc differs: {(g['jc'][m][both] != d['jc'][m][both]).sum()}")
        if jacdump:
            ctx.enable_pixel_dump(2)
            ctx.evaluate(pose, True)
            gj = ctx.pixel_dump()
            ctx.enable_pixel_dump(False)
            oj = o.dump_jac()
            for c in idf[np.argsort(-dJ)[:3]]:
                print(f"    [J] cell {c}: oracle {ref[3][c]}\n                 hip    {got[3][c]}")
                mm = (cell == c)
                co, cg = oj["jc"][mm] >= 0, gj["jc"][mm] >= 0
                print(f"    [jac dump] cell {c}: contributing pixels oracle {co.sum()} hip {cg.sum()} mismatched {np.sum(co != cg)}")
                bb = co & cg
                for key_o, key_g in (("gx", "u"), ("gy", "v"), ("pc", "ic")):
                    dd = np.abs(oj[key_o][mm][bb] - gj[key_g][mm][bb])
                    print(f"        max|d {key_o}| {np.nanmax(dd) if dd.size else 0:.3e}  (> 1e-9: {(dd > 1e-9).sum()})")
                ddw = np.abs(oj["dw"][mm][bb] - gj["wc"][mm][bb]).max(axis=1) if bb.any() else np.zeros(0)
                print(f"        max|d dw| {ddw.max() if ddw.size else 0:.3e} (> 1e-9: {(ddw > 1e-9).sum()}); jc differs {(oj['jc'][mm][bb] != gj['jc'][mm][bb]).sum()}")
                bad = np.where(ddw > 1e-9)[0][:4]
                for i in bad:
                    print("          pc oracle %.17g hip %.17g dw oracle %s hip %s" % (oj["pc"][mm][bb][i], gj["ic"][mm][bb][i], oj["dw"][mm][bb][i], gj["wc"][mm][bb][i]))
                only = np.where(co != cg)[0][:4]
                ids = np.where(mm)[0]
                for i in only:
                    print("          pixel %d (r %d c %d): oracle contributes %s, hip %s; cost u %.17g v %.17g" % (ids[i], ids[i] // pair.cols, ids[i] % pair.cols, co[i], cg[i], d["u"][ids[i]], d["v"][ids[i]]))
        k = np.where(dic > 1e-9)[0][:5]
        for i in k:
            print("      ic oracle %.17g hip %.17g  u %.17g / %.17g  v %.17g / %.17g" % (
                d["ic"][m][both][i], g["ic"][m][both][i], d["u"][m][both][i], g["u"][m][both][i], d["v"][m][both][i], g["v"][m][both][i]))


def ident_pose(pair):
    R = pair.T_wc0[:3, :3].T
    return synth.pose7_from_Rt(R, -R @ pair.T_wc0[:3, 3])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lmflash":
        nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        its = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
        flash = synth.make_pair("A", flash=True, edge_cases=True)
        o = O.from_pair(flash, nb, jac_bound="cpu", xform="matrix")
        o.compute_href(flash.pose_init)
        pose_o, recs_o = o.lm(flash.pose_init, 10)
        print("oracle trace", [(r["lm_trials"], round(r["chi2"], 6), float("%.3g" % r["rho"])) for r in recs_o])
        delta = float(np.sqrt(0.95))
        for i in its:
            r = recs_o[i]
            run(flash, nb, r["pose7"], f"flashA/matrix LM pose of iteration {r['iteration']}", "matrix")
            # the 6x6 system at that pose, both sides
            o2 = O.from_pair(flash, nb, jac_bound="cpu", xform="matrix"); o2.compute_href(flash.pose_init)
            ref = o2.evaluate(r["pose7"], True)
            Ho, bo, co, no = O.normal_equations(ref[2], ref[3], delta)
            ctx = capi.from_pair(flash, nb, xform=capi.XFORM_MATRIX); ctx.compute_href(flash.pose_init)
            H, b, c, n = ctx.normal_equations(r["pose7"], delta)
            print(f"    6x6 at iteration {i}: chi2 {c:.12f} / {co:.12f}  n {n}/{no}  max|dH|/max|H| {np.abs(H - Ho).max() / np.abs(Ho).max():.3e}  "
                  f"max|db|/max|b| {np.abs(b - bo).max() / np.abs(bo).max():.3e}  cond(H) {np.linalg.cond(Ho):.3e}  "
                  f"nonfinite cells oracle {int((~np.isfinite(ref[3]).all(axis=1) & (np.isfinite(ref[2]))).sum())}")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "seed":   # a case of tests/test_parity_gpu.py::_random_case (tools/random_parity_sweep.py)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_parity_gpu as T
        for seed in [int(x) for x in sys.argv[2:]]:
            pair, nb, poses = T._random_case(synth, 1000 + seed)
            print(f"=== seed {seed}: {pair.rows}x{pair.cols}, {pair.cell}x{pair.cell} cells, {nb} bins")
            for k, pose in enumerate(poses):
                run(pair, nb, pose, f"seed {seed} pose {k}", jacdump=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ident":
        S = synth.make_pair("S", edge_cases=True)
        run(S, 8, ident_pose(S), "S-edge identity", href_pose=ident_pose(S), jacdump=True)
        S2 = synth.make_pair("S")
        run(S2, 8, ident_pose(S2), "S identity", href_pose=ident_pose(S2))
        sys.exit(0)
    flash = synth.make_pair("A", flash=True, edge_cases=True)
    for xf in ("quat", "matrix"):
        run(flash, 8, flash.pose_init, "flashA/" + xf, xf)
    run(flash, 10, flash.pose_true, "flashA/quat")
    A = synth.make_pair("A")
    run(A, 8, A.pose_init, "A")
    S = synth.make_pair("S", edge_cases=True)
    run(S, 8, S.pose_true, "S-edge")
