import importlib, os, sys, time, json, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
for cfg in "AB":
    pair = synth.make_pair(cfg)
    t0 = time.perf_counter(); ctx = capi.from_pair(pair, 8); t1 = time.perf_counter()
    print(cfg, "context creation + uploads %.1f ms" % ((t1 - t0) * 1e3))
    ctx.compute_href(pair.pose_init)
    delta = float(np.sqrt(0.95))
    poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
    ctx.run_sequence(poses[np.arange(4096) % 256], delta, batch=64, collect=False)
    contract = ctx.contract_bytes()
    for b in (16, 32, 64, 128, 256):
        ms = float(np.median([ctx.time_launches(poses[:b], delta, repeats=6) for _ in range(9)]))
        seq = poses[np.arange(256 * 40 if cfg == "A" else 256 * 12) % 256]
        ctx.run_sequence(seq[:4 * b], delta, batch=b, collect=False)   # (ring buffers of this launch size exist now)
        t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=b, collect=False); el = time.perf_counter() - t0
        print(f"  batch {b:3d}: kernel {ms*1e3:8.1f} us = {ms*1e3/b:6.3f} us/pose, frac {contract*b/(ms*1e-3)/8e12:.3f}; pipelined {len(seq)/el:9.0f} it/s")
