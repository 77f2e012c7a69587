#!/usr/bin/env python3
"""tools/latency_sweep.py [A|B] [bins]: the latency side of the path (what a Gauss-Newton / LM loop sees) per
workgroup shape: kernel time of a single-pose cost+Jacobian launch and of cost-only launches of 1 / 4 / 10 poses,
the dependent chain (nid_run_chain: launch + kernel + result on the host), the blocking per-cell call, and the
optimize() time of the reference driver's LM on the host stack.  Output -> profiles/rNN_launch_cost_*.txt."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pair = synth.make_pair(cfg)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-3 * k, 0, 0], [0, 1e-3 * k, 0]) for k in range(16)])
ctx = capi.from_pair(pair, bins)
ctx.compute_href(pair.pose_init)
seq = poses[np.arange(4000) % 16]
ctx.run_sequence(seq, delta, batch=16, collect=False)   # clocks up
print(f"config {cfg}, {bins} bins, {pair.cell ** 2} cells; times in us (median of 9 groups of 10 back-to-back launches)")
print("threads | J 1 pose | cost 1 | cost 4 | cost 10 | chain J (launch+kernel+host) | chain cost | evaluate() per-cell")
def kt(n, jac):
    return 1e3 * float(np.median([ctx.time_launches(poses[:n], delta, repeats=10, want_jac=jac) for _ in range(9)]))
for nt in (128, 256, 512, 1024):
    ctx.set_launch_shape(nt, nt)
    row = [kt(1, True), kt(1, False), kt(4, False), kt(10, False)]
    ctx.run_chain(seq[:200], delta, want_jac=True, collect=False)
    _, el = ctx.run_chain(seq[:2000], delta, want_jac=True, collect=False)
    _, elc = ctx.run_chain(seq[:2000], delta, want_jac=False, collect=False)
    t0 = time.perf_counter()
    for i in range(300):
        ctx.evaluate(poses[i % 16], True)
    ev = (time.perf_counter() - t0) / 300
    print(f"{nt:7d} | {row[0]:8.1f} | {row[1]:6.1f} | {row[2]:6.1f} | {row[3]:7.1f} | {el / 2000 * 1e6:28.1f} | {elc / 2000 * 1e6:10.1f} | {ev * 1e6:8.1f} (python loop)")
ctx.set_launch_shape(0, 0)
print("automatic cost-only shape: cost 1 / 4 / 10 / 16 poses:", " ".join(f"{kt(n, False):.1f}" for n in (1, 4, 10, 16)))
print()
print("reference driver's LM on the host stack (10 iterations asked; optimize() only), per launch shape of the operators:")
print("jac/cost threads | per-edge flow ms | fused ms | fused + batched trials ms | outer iterations | us per outer iteration (batched)")
for jt, ct in ((256, 256), (512, 0), (1024, 0), (1024, 512), (1024, 256)):
    hostlib.set_launch_shape(jt, ct)
    t = []
    for fused in (0, 1, 2):
        hostlib.run_lm(pair, bins, pair.pose_init, 10, fused=fused)
        best = 1e9
        for _ in range(5):
            pose, recs, _ = hostlib.run_lm(pair, bins, pair.pose_init, 10, fused=fused)
            best = min(best, hostlib.last_optimize_seconds())
        t.append(best)
    print(f"{jt:5d}/{ct:<5d}      | {t[0] * 1e3:14.3f} | {t[1] * 1e3:8.3f} | {t[2] * 1e3:24.3f} | {len(recs):16d} | {t[2] / len(recs) * 1e6:8.1f}")
hostlib.set_launch_shape(512, 0)
