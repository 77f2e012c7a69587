#!/usr/bin/env python3
"""tools/shard_rate.py [A|B]: what ONE rank of an N-GPU job can evaluate per second on its share of the cells
(compute + launch path only, no exchange), per launch size -- the upper bound of the strong-scaling curve
(N x this rate vs the 1-GPU rate) and the basis of bench.py's choice of poses per launch for N > 1."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
pair = synth.make_pair(cfg)
ncell = pair.cell ** 2
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
print(f"config {cfg}: {ncell} cells; evaluations/s of rank 0 of N on its cell range (nid_run_sequence, two streams)")
print("   N | cells | batch 16 | batch 64 | batch 256 | speed-up of the evaluation rate over N = 1 (best batch)")
base = None
for n in (1, 2, 4, 8, 16, 32):
    lo, hi = capi.cell_range(0, n, ncell)
    ctx = capi.from_pair(pair, 8, cell_begin=lo, cell_end=hi)
    ctx.compute_href(pair.pose_init)
    row = []
    for b in (16, 64, 256):
        seq = poses[np.arange(256 * 60) % 256]
        ctx.run_sequence(seq[:256 * 8], delta, batch=b, collect=False)
        t0 = time.perf_counter()
        ctx.run_sequence(seq, delta, batch=b, collect=False)
        row.append(len(seq) / (time.perf_counter() - t0))
    if base is None:
        base = max(row)
    print(f"{n:4d} | {hi - lo:5d} | {row[0]:8.0f} | {row[1]:8.0f} | {row[2]:8.0f} | {max(row) / base:6.2f}")
