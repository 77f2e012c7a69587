# ARCHIVED: round 4 probe: per-step stamps of a batch launch; kept because profiles/ and HISTORY.md cite its output (as tools/stamps_batch.py). Not part of the test or measurement flow.
"""Diagnostic: per-phase cycle shares of the evaluation kernel inside a BATCHED launch
(stamps are taken by the workgroups of pose 0; the other poses provide the contention).
Usage: python tools/stamps_batch.py [A|B] [bins] [batch]"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pair = synth.make_pair(cfg)
ctx = capi.from_pair(pair, bins)
cnt, _ = ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(batch)])
for _ in range(5):
    ctx.launch_batch(0, poses, delta)
    for k in range(batch): ctx.wait(k)
ctx.enable_stamps(True)
for rep in range(3):
    ctx.launch_batch(0, poses, delta)
    for k in range(batch): ctx.wait(k)
    st = ctx.stamps()
    act = cnt >= 300
    s = st[act]
    names = ["zero+tables", "phase1(cost)", "fold+entropy", "phase2(jac)", "blocksum6", "cellquad", "reduce tail"]
    d = np.diff(s[:, :8], axis=1)
    rt = (s[:, 9] - s[:, 8]) / 100.0
    cyc = (s[:, 7] - s[:, 0])
    print("rep", rep, "clock est %.3f GHz; block time median %.2f us max %.2f us" % (np.median(cyc / rt) / 1e3, np.median(rt), rt.max()))
    for k, n in enumerate(names):
        print(f"  {n:16s} median {np.median(d[:,k]):9.0f} cyc   p90 {np.percentile(d[:,k],90):9.0f}  max {d[:,k].max():9.0f}")
    print("  span of pose-0 blocks (wall): %.2f us; start spread %.2f us" % ((s[:, 9].max() - s[:, 8].min()) / 100.0, (s[:, 8].max() - s[:, 8].min()) / 100.0))
