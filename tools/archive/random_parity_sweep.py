#!/usr/bin/env python3
# ARCHIVED: the single-process random sweep of rounds 2-3 (superseded by tools/parity_sweep_mp.py); output recorded in / cited by: profiles/r02_ablations_A.txt, HISTORY.md, tests/test_parity_gpu.py (there as tools/random_parity_sweep.py). Not part of the test or measurement flow.
"""tools/random_parity_sweep.py FIRST COUNT [shape]: tests/test_parity_gpu.py::test_randomised_pairs over seeds beyond the
64 the suite runs (a one-off sweep on the GPU box: the exact-decision / fine-level machinery of FAST math has many rare
corners).  Also evaluates every case in the latency launch shape when `shape` (512 / 1024) is given.  Prints the seeds
that violate the suite's tolerances; exit code 1 if any."""
import importlib, os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as oracle
import test_parity_gpu as T
first, count = int(sys.argv[1]), int(sys.argv[2])
shape = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = []
for seed in range(first, first + count):
    try:
        pair, nb, poses = T._random_case(synth, 1000 + seed)
        o = oracle.from_pair(pair, nb)
        cnt_o, href_o = o.compute_href(pair.pose_init)
        act = cnt_o >= 300
        refs = [o.evaluate(p, True) for p in poses]
        for math in T.MODES:
            ctx = capi.from_pair(pair, nb, math=T._mode(capi, math))
            if shape:
                ctx.set_launch_shape(shape, shape)
            cnt, href = ctx.compute_href(pair.pose_init)
            assert np.array_equal(cnt, cnt_o) and np.array_equal(np.isnan(href), ~act)
            np.testing.assert_allclose(href[act], href_o[act], rtol=0, atol=T.ATOL_H)
            for pose, ref in zip(poses, refs):
                T._compare_cells(ctx.evaluate(pose, True), ref, cnt_o, noise=(o, pose))
                assert ctx.normal_equations(pose, T.DELTA)[3] == int(act.sum())
            ctx.close()
    except Exception as e:   # noqa: BLE001
        bad.append(seed)
        print(f"seed {seed}: {type(e).__name__}: {str(e)[:300]}")
print(f"seeds {first}..{first + count - 1}" + (f" at {shape} threads" if shape else "") + f": {count - len(bad)} ok, {len(bad)} failing {bad}")
sys.exit(1 if bad else 0)
