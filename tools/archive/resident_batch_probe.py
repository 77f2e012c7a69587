#!/usr/bin/env python3
"""tools/resident_batch_probe.py [config]: microseconds per request of K poses (cost + Jacobian and cost-only) -- launched
(nid_launch_batch: plan_split's launches) against the resident batch evaluator (nid_set_resident) -- in a loop of
launch_batch + wait of every slot, 300 requests each, median of 5 blocks.  profiles/r05_short_sequences.txt."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
pair = synth.make_pair(cfg)
ctx = capi.from_pair(pair, 8)
ctx.compute_href(pair.pose_init)
rng = np.random.default_rng(5)
poses = np.stack([synth.perturb_pose7(pair.pose_init, rng.normal(0, 1.5e-3, 3), rng.normal(0, 2e-3, 3)) for _ in range(64)])
delta = float(np.sqrt(0.95))
lib, h = ctx.lib, ctx.h
import ctypes as C


def shot(K, jac):
    p = np.ascontiguousarray(poses[:K])
    out = np.zeros((K, 32))
    def once():
        rc = lib.nid_run_sequence(h, p.ctypes.data_as(C.POINTER(C.c_double)), K, 256, 1 if jac else 0, delta, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0
    for _ in range(30):
        once()
    blocks = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(300):
            once()
        blocks.append((time.perf_counter() - t0) / 300 * 1e6)
    return float(np.median(blocks))


print(f"# config {cfg}: us per request of K poses (nid_run_sequence, n <= batch), launched | resident batch evaluator")
for jac in (True, False):
    for K in (1, 2, 4, 8, 10, 16, 20, 24, 32, 64):
        ctx.set_resident(False)
        a = shot(K, jac)
        ctx.set_resident(2)
        b = shot(K, jac)
        st = ctx.resident_batch_stats()
        print(f"{'cost+Jacobian' if jac else 'cost only    '} K {K:3d}: launched {a:7.1f} us ({K / a * 1e6 / 1e3:6.1f} k/s)   resident {b:7.1f} us ({K / b * 1e6 / 1e3:6.1f} k/s)   served {st['served']} fallbacks {st['fallbacks']}")
        ctx.set_resident(False)
