import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
pair = synth.make_pair(cfg)
ncell = pair.cell ** 2
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
interleaved = len(sys.argv) > 2 and sys.argv[2] == "interleaved"
for n in (2, 4, 8):
    rates = []
    for k in range(n):
        if interleaved:
            lo, hi = k, ncell
            ctx = capi.from_pair(pair, 8, cell_begin=k, cell_end=ncell, cell_stride=n)
            own = np.arange(k, ncell, n)
        else:
            lo, hi = capi.cell_range(k, n, ncell)
            ctx = capi.from_pair(pair, 8, cell_begin=lo, cell_end=hi)
            own = np.arange(lo, hi)
        cnt, _ = ctx.compute_href(pair.pose_init)
        seq = poses[np.arange(256 * 40) % 256]
        ctx.run_sequence(seq[:256 * 8], delta, batch=256, collect=False)
        t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=256, collect=False)
        rates.append((len(seq) / (time.perf_counter() - t0), int((cnt[own] >= 300).sum())))
        ctx.close()
    print(cfg, "interleaved" if interleaved else "contiguous", "N =", n, " per-rank rate (k it/s) / active cells:", " ".join(f"{r/1e3:.0f}/{a}" for r, a in rates), " -> job rate = min = %.0f k" % (min(r for r, _ in rates) / 1e3))
