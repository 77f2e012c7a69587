#!/usr/bin/env python3
"""tools/rare_census.py: how full the second pass's rounds are (census build: tools/build_variant.py census -DNID_CENSUS, loaded from
exp/libnid_census.so): per wave of a 256-pose cost + Jacobian launch of the 128-thread kernel (config A, 8 bins), the rare rounds it
runs now (R), the rare samples in them and the rounds the same samples would take packed 64 to a round -- plain and flash pair."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NID_HIP_LIB", os.path.join(ROOT, "exp", "libnid_census.so"))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
lib = ctypes.CDLL(os.environ["NID_HIP_LIB"])
delta = float(np.sqrt(0.95))
for name, kw in (("plain", {}), ("flash", dict(flash=True, edge_cases=True))):
    pair = synth.make_pair("A", **kw)
    ctx = capi.from_pair(pair, int(os.environ.get("NID_AB_BINS", "8")))
    ctx.compute_href(pair.pose_init)
    poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
    out = (ctypes.c_ulonglong * 64)()
    ctx.run_sequence(poses, delta, batch=256, collect=False)
    assert lib.nid_census_read(out, 1) == 0
    ctx.run_sequence(poses, delta, batch=256, collect=False)
    assert lib.nid_census_read(out, 1) == 0
    c = np.array(out[:], dtype=np.int64)
    waves = 256 * 256 * 2
    print(f"{name}: {waves} waves per launch, {c[0]} with a second pass ({100 * c[0] / waves:.1f} %); rare rounds now {c[1]} ({c[1] / waves:.2f} per wave), "
          f"packed per wave {c[2]} ({c[2] / waves:.2f}); rare samples {c[3]} = {c[3] / max(c[1], 1):.1f} per round now, {c[3] / max(c[2], 1):.1f} packed")
    print("   waves by rounds now    R = 1..:", " ".join(str(v) for v in c[9:9 + 12]), " (R >= 13:", int(c[21:40].sum()), ")")
    print("   waves by rounds packed   1..:", " ".join(str(v) for v in c[41:41 + 8]))
    ctx.close()
