#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; kept because profiles/ and HISTORY.md cite its output (as tools/diag_forms.py). Not part of the test or measurement flow.
"""tools/diag_forms.py NB NT: latency form vs loop form, cell by cell (tests/test_parity_gpu.py::test_latency_form_equals_loop_form)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
import test_parity_gpu as T
nb, nt = int(sys.argv[1]), int(sys.argv[2])
for name, pair in (("A", synth.make_pair("A")), ("A flash", synth.make_pair("A", flash=True)), ("S", synth.make_pair("S"))):
    ctx = capi.from_pair(pair, nb)
    ctx.set_launch_shape(nt, nt)
    ctx.compute_href(pair.pose_init)
    poses = list(T._poses(synth, pair).values()) + [T._identity_pose(synth, pair)]
    for pi, p in enumerate(poses):
        res = {}
        for loop in (False, True):
            ctx.set_loop_form(loop)
            ctx.repair_count(reset=True)
            out = [ctx.evaluate(p, True) for _ in range(3)]
            res[loop] = (out, ctx.repair_count())
        a, b = res[False][0][0], res[True][0][0]
        if os.environ.get("DIAG_ORACLE") and name == "A flash" and pi == 0:
            from oracle import oracle_py
            o = oracle_py.from_pair(pair, nb)
            o.compute_href(pair.pose_init)
            ref = o.evaluate(p, True)
            for f, nm2 in ((False, "lat"), (True, "loop")):
                J = res[f][0][0][3]
                act = np.isfinite(ref[3]).all(axis=1) & np.isfinite(J).all(axis=1)
                rel = np.abs(J[act] - ref[3][act]).max(axis=1) / np.abs(ref[3][act]).max(axis=1)
                bad = np.where(act)[0][rel > 1e-9]
                print(f"   vs oracle, {nm2}: worst {rel.max():.2e}; cells over 1e-9: {bad[:12]}")
        for k, nm in enumerate(("Hc", "Hj", "err", "J")):
            x, y = np.atleast_2d(T._bits(a[k])), np.atleast_2d(T._bits(b[k]))
            d = np.where((x != y).reshape(x.shape[0] if k == 3 else -1, -1).any(axis=1))[0] if k == 3 else np.where(T._bits(a[k]) != T._bits(b[k]))[0]
            if len(d):
                print(f"{name} pose {pi} {nm}: {len(d)} cells differ {d[:8]}; lat {np.asarray(a[k])[d[0]]} loop {np.asarray(b[k])[d[0]]}")
        rr = [[np.array_equal(T._bits(res[f][0][0][k]), T._bits(res[f][0][r][k])) for k in range(4) for r in (1, 2)] for f in (False, True)]
        print(f"{name} pose {pi}: repairs lat {res[False][1]} loop {res[True][1]}; run-to-run same: lat {all(rr[0])} loop {all(rr[1])}")
