#!/usr/bin/env python3
"""tools/first_shot_probe.py: why bench.py's ONE timed 20-pose shot (--steps 20 --warmup 5) is slower than the same shot repeated.
Replays bench.py's order -- preheat on run(20, collect=False), run(5), torch.cuda.synchronize(), the timed run(20) -- in
variants, ten times each, us per shot (median, min .. max)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair("A")
ctx = capi.from_pair(pair, 8)
ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
dev = torch.device("cuda:0")
seq20 = np.ascontiguousarray(poses[:20]); seq5 = np.ascontiguousarray(poses[:5])


def preheat(seconds, with_torch, iters=100000):
    t = time.perf_counter()
    if seconds <= 0 and iters == 100000: return
    for _ in range(iters):
        ctx.run_sequence(seq20, delta, batch=256, collect=False)
        if with_torch:
            go_on = torch.tensor([1.0 if time.perf_counter() - t < seconds else 0.0], dtype=torch.float64)
            if go_on.item() == 0.0: break
        elif time.perf_counter() - t >= seconds: break


def shot(pre_s, with_torch, warm_collect, gap_s, iters=100000):
    preheat(pre_s, with_torch, iters)
    if warm_collect: ctx.run_sequence(seq20, delta, batch=256, collect=True)
    ctx.run_sequence(seq5, delta, batch=256, collect=False)
    torch.cuda.synchronize(dev)
    if gap_s: time.sleep(gap_s)
    t0 = time.perf_counter(); ctx.run_sequence(seq20, delta, batch=256, collect=True); t1 = time.perf_counter()
    reps = []
    for _ in range(3):
        ctx.run_sequence(seq5, delta, batch=256, collect=False)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter(); ctx.run_sequence(seq20, delta, batch=256, collect=True); reps.append((time.perf_counter() - t2) * 1e6)
    return (t1 - t0) * 1e6, reps


for label, kw in (("bench's order (preheat 0.25 s with the torch flag)", dict(pre_s=0.25, with_torch=True, warm_collect=False, gap_s=0)),
                  ("preheat without the torch flag", dict(pre_s=0.25, with_torch=False, warm_collect=False, gap_s=0)),
                  ("one collecting run(20) before the warmup steps", dict(pre_s=0.25, with_torch=True, warm_collect=True, gap_s=0)),
                  ("no preheat", dict(pre_s=0.0, with_torch=False, warm_collect=False, gap_s=0)),
                  ("1 ms of sleep between the synchronisation and the shot", dict(pre_s=0.25, with_torch=True, warm_collect=False, gap_s=0.001)),
                  ("50 ms of sleep between the synchronisation and the shot", dict(pre_s=0.25, with_torch=True, warm_collect=False, gap_s=0.05)),
                  ("preheat of 5 sequences", dict(pre_s=10.0, with_torch=False, warm_collect=False, gap_s=0, iters=5)),
                  ("preheat of 20 sequences", dict(pre_s=10.0, with_torch=False, warm_collect=False, gap_s=0, iters=20)),
                  ("preheat of 100 sequences", dict(pre_s=10.0, with_torch=False, warm_collect=False, gap_s=0, iters=100)),
                  ("preheat of 500 sequences", dict(pre_s=10.0, with_torch=False, warm_collect=False, gap_s=0, iters=500)),
                  ("preheat of 2000 sequences", dict(pre_s=10.0, with_torch=False, warm_collect=False, gap_s=0, iters=2000)),
                  ("no preheat (again)", dict(pre_s=0.0, with_torch=False, warm_collect=False, gap_s=0))):
    first, reps = [], []
    for _ in range(10):
        f, r = shot(**kw)
        first.append(f); reps.append(r)
    first = np.array(first); reps = np.array(reps)
    print(f"{label:60s} first shot median {np.median(first):6.1f} us ({first.min():6.1f} .. {first.max():6.1f});  repeats 1..3 medians "
          + " ".join(f"{np.median(reps[:, i]):6.1f}" for i in range(3)))
