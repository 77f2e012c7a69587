#!/usr/bin/env python3
"""tools/batch_quantum.py: poses per launch against the chip's workgroup slots on a SMALL shard (one rank of an N-GPU job).  The
blockIdx -> (cell, pose) map deals cells to the 8 XCDs by cell index; an XCD has 32 CUs x 10 resident workgroups of the 128-thread
kernel = 320 slots, so a launch of c cells per XCD x B poses runs in ceil(c * B / 320) rounds: 4 cells x 256 poses = 3.2 -> 4 rounds
(80 %), 4 x 240 = 3.0.  Evaluations/s of rank 0's interleaved share per batch size (nid_run_sequence, two streams)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
pair = synth.make_pair(cfg)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
batches = (256, 250, 240, 224, 200, 192, 160, 128)
print(f"config {cfg}: evaluations/s (k) of rank 0 of N, interleaved cells, by poses per launch")
print("   N | cells (active) | " + " | ".join(f"{b:6d}" for b in batches))
for n in (1, 2, 4, 8):
    ctx = capi.from_pair(pair, 8, cell_begin=0, cell_stride=n)
    cnt, _ = ctx.compute_href(pair.pose_init)
    row = []
    for b in batches:
        seq = poses[np.arange(b * 160) % 256]
        ctx.run_sequence(seq[:b * 16], delta, batch=b, collect=False)
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=b, collect=False); best = max(best, len(seq) / (time.perf_counter() - t0))
        row.append(best)
    print(f"  {n:2d} | {len(cnt):4d} ({int((cnt >= 300).sum()):4d})    | " + " | ".join(f"{r / 1e3:6.0f}" for r in row), flush=True)
    ctx.close()
