#!/usr/bin/env python3
"""tools/direct_latency.py [A|B] [bins]: the dependent evaluation (one pose per launch, result awaited on the host before
the next launch: what a Gauss-Newton / LM loop does) with DIRECT results (every cell's block straight to pinned host
memory, summed by the host), GROUP-DIRECT results (the groups' sums formed on the device, added up by the host) and the
in-launch two-level reduction, per workgroup shape; us per evaluation."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
pair = synth.make_pair(cfg)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-3 * k, 0, 0], [0, 1e-3 * k, 0]) for k in range(16)])
ctx = capi.from_pair(pair, bins)
ctx.compute_href(pair.pose_init)
seq = poses[np.arange(4000) % 16]
ctx.run_sequence(seq, delta, batch=16, collect=False)   # clocks up
print(f"config {cfg}, {bins} bins, {pair.cell ** 2} cells: us per dependent evaluation (nid_run_chain, best of 5 runs of 2000)")
print("threads | J in-launch | J group-direct | J direct | J resident | cost in-launch | cost group-direct | cost direct | cost resident | evaluate() in-launch | direct | resident (python loop)")
resident_why = ""


def set_resident(on):
    """nid_set_resident, tolerating a platform (or geometry) without it: the column then shows launches"""
    global resident_why
    try:
        ctx.set_resident(on)
    except capi.NidError as e:
        resident_why = resident_why or str(e)   # (the first refusal says why)


for nt in (128, 256, 512, 1024):
    ctx.set_launch_shape(nt, nt)
    row = []
    modes = ((False, False), (True, False), (True, True))   # (direct, resident)
    for jac in (True, False):
        for direct, resident in ((0, False), (2, False), (1, False), (1, True)):
            ctx.set_direct_results(direct)
            set_resident(resident and nt in (256, 512))
            ctx.run_chain(seq[:200], delta, want_jac=jac, collect=False)
            row.append(min(ctx.run_chain(seq[:2000], delta, want_jac=jac, collect=False)[1] for _ in range(5)) / 2000 * 1e6)
    ev = []
    for direct, resident in modes:
        ctx.set_direct_results(direct)
        set_resident(resident and nt in (256, 512))
        t0 = time.perf_counter()
        for i in range(300):
            ctx.evaluate(poses[i % 16], True)
        ev.append((time.perf_counter() - t0) / 300 * 1e6)
    ctx.set_resident(False)
    print(f"{nt:7d} | {row[0]:11.1f} | {row[1]:14.1f} | {row[2]:8.1f} | {row[3]:10.1f} | {row[4]:14.1f} | {row[5]:17.1f} | {row[6]:11.1f} | {row[7]:13.1f} | {ev[0]:20.1f} | {ev[1]:6.1f} | {ev[2]:8.1f}")
print("resident evaluator:", ctx.resident_stats(), resident_why)
