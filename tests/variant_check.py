"""Quick validity gate for an experiment build (NID_HIP_LIB=exp/libnid_X.so python tests/variant_check.py): config S and A evaluations against the
CPU oracle at the FAST-mode tolerances.  Exit code 1 on mismatch."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py
ok = True
for cfg, nb in (("S", 8), ("A", 8), ("S", 10)):
    pair = synth.make_pair(cfg)
    ctx = capi.from_pair(pair, nb)
    o = oracle_py.from_pair(pair, nb)
    cnt, href = ctx.compute_href(pair.pose_init)
    o.compute_href(pair.pose_init)
    Hc, Hj, err, J = ctx.evaluate(pair.pose_init, True)
    Hc_o, Hj_o, err_o, J_o = o.evaluate(pair.pose_init, True)
    act = cnt >= 300
    dH = max(np.nanmax(np.abs(Hc[act] - Hc_o[act])), np.nanmax(np.abs(Hj[act] - Hj_o[act])))
    scale = np.maximum(1.0, np.abs(J_o[act]).max(axis=1, keepdims=True))
    dJ = np.nanmax(np.abs(J[act] - J_o[act]) / scale)
    good = dH < 1e-4 and dJ < 5e-2   # loose gate (saturated cells, see tests/test_parity_gpu.py); typical 1e-13
    print(f"{cfg} nb={nb}: max|dH| {dH:.2e}  max|dJ|/scale {dJ:.2e}  {'ok' if good else 'MISMATCH'}")
    ok &= bool(good)
sys.exit(0 if ok else 1)
