# ARCHIVED: round 2 probe of launch overheads; output recorded in / cited by: profiles/README.md (there as tools/launch_cost.py). Not part of the test or measurement flow.
"""Diagnostic: host cost of a launch call and device time of batched launches."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair(sys.argv[1] if len(sys.argv) > 1 else "A")
ctx = capi.from_pair(pair, 8)
ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-3 * k, 0, 0], [0, 1e-3 * k, 0]) for k in range(8)])
for _ in range(20):
    ctx.launch(0, poses[0], delta); ctx.wait(0)
# host cost of launch calls (queue them, no waiting in between)
for B in (1, 2, 4, 8):
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        s = (i * B) % 16
        if i * B >= 16:
            for k in range(B): ctx.wait((s + k) % 16)
        if B == 1: ctx.launch(s, poses[0], delta)
        else: ctx.launch_batch(s, poses[:B], delta)
    t1 = time.perf_counter()
    for k in range(16):
        try: ctx.wait(k)
        except Exception: pass
    t2 = time.perf_counter()
    ctx.enable_timing(True)
    ms = []
    for i in range(30):
        if B == 1: ctx.launch(0, poses[0], delta)
        else: ctx.launch_batch(0, poses[:B], delta)
        for k in range(B): ctx.wait(k)
        ms.append(ctx.last_kernel_ms(0)[0])
    ctx.enable_timing(False)
    # blocking latency of one launch+wait
    t3 = time.perf_counter()
    for i in range(100):
        if B == 1: ctx.launch(0, poses[0], delta)
        else: ctx.launch_batch(0, poses[:B], delta)
        for k in range(B): ctx.wait(k)
    t4 = time.perf_counter()
    print(f"B={B}: pipelined {1e6*(t2-t0)/n/B:.1f} us/pose (enqueue loop {1e6*(t1-t0)/n:.1f} us/launch), "
          f"kernel {1e3*np.median(ms):.1f} us/launch = {1e3*np.median(ms)/B:.1f} us/pose, "
          f"launch+wait latency {1e6*(t4-t3)/100:.1f} us")
