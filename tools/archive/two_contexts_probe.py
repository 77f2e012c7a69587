#!/usr/bin/env python3
"""tools/archive/two_contexts_probe.py: would MORE than two streams help the pipelined loop on a small shard (one rank of an
N-GPU job)?  Two contexts of the same 1/N of the cells (interleaved), each running nid_run_sequence on its own two streams from
its own host thread (ctypes releases the GIL), against one context alone: evaluations/s in total (round 6)."""
import importlib, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair("A")
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
seq = poses[np.arange(256 * 120) % 256]
for n in (1, 2, 4, 8, 16):
    ctxs = [capi.from_pair(pair, 8, cell_begin=0, cell_stride=n) for _ in range(3)]
    for c in ctxs:
        c.compute_href(pair.pose_init)
        c.run_sequence(seq[:256 * 20], delta, batch=256, collect=False)
    def rate(k):
        th = [threading.Thread(target=lambda c=c: c.run_sequence(seq, delta, batch=256, collect=False)) for c in ctxs[:k]]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        return k * len(seq) / (time.perf_counter() - t0)
    r1 = max(rate(1) for _ in range(3)); r2 = max(rate(2) for _ in range(3)); r3 = max(rate(3) for _ in range(3))
    print(f"N = {n:2d} (every {n}th cell): one context {r1 / 1e3:8.0f} k it/s, two at once {r2 / 1e3:8.0f} k ({r2 / r1:.2f}x), three at once {r3 / 1e3:8.0f} k ({r3 / r1:.2f}x)", flush=True)
    for c in ctxs: c.close()
