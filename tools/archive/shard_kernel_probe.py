import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair("A")
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
seq = poses[np.arange(256 * 160) % 256]
for n in (1, 2, 4, 8, 16):
    ctx = capi.from_pair(pair, 8, cell_begin=0, cell_stride=n)
    ctx.compute_href(pair.pose_init)
    ctx.run_sequence(seq[:256 * 16], delta, batch=256, collect=False)
    t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=256, collect=False); el = time.perf_counter() - t0
    k = np.median([ctx.time_kernel(poses, delta, repeats=10) for _ in range(5)]) * 1e3
    l = np.median([ctx.time_launches(poses, delta, repeats=10) for _ in range(5)]) * 1e3
    print(f"N = {n:2d}: pipelined {el / 160 * 1e6:7.1f} us per launch ({len(seq) / el / 1e3:6.0f} k it/s); k_eval2 alone {k:7.1f} us; a launch alone {l:7.1f} us; ideal N-th of N = 1's kernel", flush=True)
    ctx.close()
