#!/usr/bin/env python3
"""tools/timed_region_probe.py: what bench.py's timed region costs at the driver's flags (--steps 20): the 20-pose sequence
through the C-ABI, the Python wrapper around it, and torch.cuda.synchronize() behind it (us, medians of 200)."""
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
for K in (20, 64, 256):
    seq = np.ascontiguousarray(poses[np.arange(K) % 256])
    for _ in range(300): ctx.run_sequence(seq, delta, batch=256)
    a, b, c = [], [], []
    for _ in range(200):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); r = ctx.run_sequence(seq, delta, batch=256); t1 = time.perf_counter()
        torch.cuda.synchronize(dev); t2 = time.perf_counter()
        torch.cuda.synchronize(dev); t3 = time.perf_counter()
        a.append(t1 - t0); b.append(t2 - t1); c.append(t3 - t2)
    med = lambda x: float(np.median(x)) * 1e6
    print(f"K {K:4d}: run_sequence {med(a):7.1f} us, synchronize behind it {med(b):6.1f} us, a second synchronize {med(c):6.1f} us  -> {K / (med(a) + med(b)) * 1e3:.0f} k it/s in a timed region")
# bench.py's own order: preheat on the timed length, the driver's warmup (another length), barrier, ONE timed shot
seq20 = np.ascontiguousarray(poses[np.arange(20) % 256]); seq5 = np.ascontiguousarray(poses[:5])
for label, warm in (("warmup 5 poses before the shot", seq5), ("warmup 20 poses before the shot", seq20), ("no warmup call", None)):
    shots = []
    for _ in range(40):
        for _ in range(50): ctx.run_sequence(seq20, delta, batch=256, collect=False)
        if warm is not None: ctx.run_sequence(warm, delta, batch=256, collect=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); r = ctx.run_sequence(seq20, delta, batch=256); torch.cuda.synchronize(dev); t1 = time.perf_counter()
        shots.append((t1 - t0) * 1e6)
    shots = np.array(shots)
    print(f"single shots, {label}: median {np.median(shots):.1f} us, min {shots.min():.1f}, max {shots.max():.1f}")
# does a device that is not idle take the first launch faster?  A second, small context keeps its resident evaluator on 16 CUs
# (one request every now and then keeps it from retiring) while the same single shots are taken
if os.environ.get("PROBE_KEEPALIVE"):
    small = synth.make_pair("S")
    c2 = capi.from_pair(small, 8); c2.compute_href(small.pose_init); c2.set_launch_shape(512, 0)
    try:
        c2.set_resident(True)
        c2.normal_equations(small.pose_init, delta)
        shots = []
        for _ in range(40):
            for _ in range(50): ctx.run_sequence(seq20, delta, batch=256, collect=False)
            c2.normal_equations(small.pose_init, delta)   # (keeps the resident kernel alive)
            torch.cuda.synchronize(dev) if False else None  # a device-wide synchronize would wait for the resident kernel: the streams instead
            t0 = time.perf_counter(); r = ctx.run_sequence(seq20, delta, batch=256); t1 = time.perf_counter()
            shots.append((t1 - t0) * 1e6)
            time.sleep(0.002)
        shots = np.array(shots)
        print(f"single shots 2 ms after the last work, another context's resident kernel on the device: median {np.median(shots):.1f} us, min {shots.min():.1f}, max {shots.max():.1f}  {c2.resident_stats()}")
        c2.set_resident(False)
        shots = []
        for _ in range(40):
            for _ in range(50): ctx.run_sequence(seq20, delta, batch=256, collect=False)
            t0 = time.perf_counter(); r = ctx.run_sequence(seq20, delta, batch=256); t1 = time.perf_counter()
            shots.append((t1 - t0) * 1e6)
            time.sleep(0.002)
        shots = np.array(shots)
        print(f"single shots 2 ms after the last work, idle device:                                      median {np.median(shots):.1f} us, min {shots.min():.1f}, max {shots.max():.1f}")
    except capi.NidError as e:
        print("keep-alive probe skipped:", e)
