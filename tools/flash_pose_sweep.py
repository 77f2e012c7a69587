#!/usr/bin/env python3
"""tools/flash_pose_sweep.py COUNT: per-cell parity (the suite's tolerances) on the 640x480 FLASH pair (saturated hot
spot, black / saturated patches, depth holes) at COUNT random poses around the initial one, 8 / 10 / 16 bins, both math
modes and the 128- and 512-thread launch shapes.  One-off sweep on the GPU box; exit code 1 on any violation."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as oracle
import test_parity_gpu as T
count = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pair = synth.make_pair("A", flash=True, edge_cases=True)
rng = np.random.default_rng(2024)
bad = 0
for nb in (8, 10, 16):
    o = oracle.from_pair(pair, nb)
    cnt_o, _ = o.compute_href(pair.pose_init)
    ctxs = []
    for math in T.MODES:
        for shape in (128, 512):
            c = capi.from_pair(pair, nb, math=T._mode(capi, math))
            c.set_launch_shape(shape, shape)
            c.compute_href(pair.pose_init)
            ctxs.append((math, shape, c))
    for i in range(count):
        scale = [1e-4, 3e-3, 3e-2][i % 3]
        pose = synth.perturb_pose7(pair.pose_init, rng.normal(0, scale, 3), rng.normal(0, 2 * scale, 3))
        ref = o.evaluate(pose, True)
        for math, shape, c in ctxs:
            try:
                T._compare_cells(c.evaluate(pose, True), ref, cnt_o, noise=(o, pose))
            except AssertionError as e:
                bad += 1
                print(f"nb {nb} pose {i} (scale {scale}) {math} {shape}: {str(e)[:260]}")
    print(f"nb {nb}: {count} poses x {len(ctxs)} variants done, violations so far {bad}")
sys.exit(1 if bad else 0)
