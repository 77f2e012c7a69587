#!/usr/bin/env python3
"""tools/classify_noise_seeds.py SEED...: what kind of case a sweep's 'needed_reference_noise' entry is -- no GPU: the case is
regenerated (tests/test_parity_gpu.py::_random_case) and evaluated by the oracle; printed are the number of grey levels of
the TARGET image (1 = a constant image: the reference's Jacobian there is its own bilinear / central-difference rounding
noise, the HIP path returns exact zeros) and the smallest per-cell Jacobian scale over the case's poses."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
import test_parity_gpu as T
O.load()
for seed in (int(x) for x in sys.argv[1:]):
    pair, nb, poses = T._random_case(synth, 1000 + seed)
    o = O.from_pair(pair, nb); cnt, _ = o.compute_href(pair.pose_init); act = cnt >= 300
    mins = []
    for p in poses:
        J = o.evaluate(p, True)[3]
        fin = np.isfinite(J).all(axis=1) & act
        if fin.any(): mins.append(float(np.abs(J[fin]).max(axis=1).min()))
    print(f"seed {seed}: {pair.rows}x{pair.cols}, {nb} bins: grey levels in the target {len(np.unique(pair.im1))}; "
          f"smallest cell Jacobian scale over the poses {min(mins) if mins else float('nan'):.2e}")
