"""Diagnostic: per-phase cycle shares of the evaluation kernel (s_memtime stamps).
Usage: python tools_stamps.py [A|B] [bins] [block_threads]"""
import importlib, sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pair = synth.make_pair(cfg)
ctx = capi.from_pair(pair, bins)
if nt: ctx.set_block_threads(nt)
cnt, _ = ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
for _ in range(5):
    ctx.normal_equations(pair.pose_init, delta)
ctx.enable_stamps(True)
for _ in range(6):   # the diagnostic instantiation's code must be warm in the instruction caches too
    ctx.normal_equations(pair.pose_init, delta)
st = ctx.stamps()
act = cnt >= 300
s = st[act]
names = ["zero+loads+warp+sample", "bspline+atomics", "fold+entropy", "phase2(jac)", "blocksum12", "cellquad", "reduce tail"]
d = np.diff(s[:, :8], axis=1)
rt = (s[:, 9] - s[:, 8]) / 100.0   # us (s_memrealtime ticks at 100 MHz)
cyc = (s[:, 7] - s[:, 0])
print("in-kernel clock estimate: median %.3f GHz (block time median %.2f us)" % (np.median(cyc / rt) / 1e3, np.median(rt)))
print("cfg", cfg, "bins", bins, "nt", nt, "active", act.sum())
for k, n in enumerate(names):
    print(f"{n:26s} median {np.median(d[:,k]):9.0f} cyc   max {d[:,k].max():9.0f}")
print("total per block median", np.median(s[:,7]-s[:,0]), "max", (s[:,7]-s[:,0]).max())
# s_memtime counters differ between XCDs: spans across workgroups from s_memrealtime (100 MHz, chip-wide)
print("kernel span (first workgroup start -> last workgroup end) %.2f us" % ((s[:, 9].max() - s[:, 8].min()) / 100.0))
print("start spread %.2f us, end spread %.2f us" % ((s[:, 8].max() - s[:, 8].min()) / 100.0, (s[:, 9].max() - s[:, 9].min()) / 100.0))
