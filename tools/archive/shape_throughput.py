#!/usr/bin/env python3
# ARCHIVED: round 2-3 probe: throughput by workgroup shape; kept because profiles/ and HISTORY.md cite its output (as tools/shape_throughput.py). Not part of the test or measurement flow.
"""tools/shape_throughput.py [A|B]: 256-pose cost + Jacobian launches (one at a time: kernel us) and the pipelined rate per
workgroup shape of the throughput path (128 = default, 256)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
pair = synth.make_pair(cfg)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
seq = poses[np.arange(256 * 60) % 256]
for nt in (128, 256, 128, 256):
    ctx = capi.from_pair(pair, 8)
    ctx.compute_href(pair.pose_init)
    ctx.set_launch_shape(nt, nt)
    ctx.run_sequence(seq[:256 * 20], delta, batch=256, collect=False)
    t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=256, collect=False); el = time.perf_counter() - t0
    ms = float(np.median([ctx.time_launches(poses, delta, repeats=10) for _ in range(7)]))
    print(f"config {cfg}, {nt} threads: kernel {ms * 1e3:.1f} us per 256 poses, pipelined {len(seq) / el:.0f} it/s")
    ctx.close()
