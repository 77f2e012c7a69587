#!/usr/bin/env python3
"""tools/soak.py [A|B] [seconds]: the pipelined path under sustained load -- every collected 6x6 block must be the
bits of the first block computed for the same pose (256 poses cycling), on one context (256 poses per launch, four
launches in flight), on 3 shards of one GPU (host sum) and through a one-rank RCCL communicator."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
pair = synth.make_pair(cfg)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
bits = lambda a: np.ascontiguousarray(a).view(np.uint64)

def soak(name, runner, n_chunk):
    ref = None
    total, bad, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        seq = poses[np.arange(n_chunk) % 256]
        out = runner(seq)
        if ref is None:
            ref = out[:256].copy()
        want = ref[np.arange(n_chunk) % 256]
        bad += int((bits(out) != bits(want)).any(axis=1).sum())
        total += n_chunk
    el = time.perf_counter() - t0
    print(f"{name:28s} {total:9d} evaluations in {el:5.1f} s ({total / el / 1e3:6.1f} k/s incl. checking), mismatching blocks: {bad}")
    return bad

ctx = capi.from_pair(pair, 8)
ctx.compute_href(pair.pose_init)
bad = soak("one context, batch 256", lambda s: ctx.run_sequence(s, delta, batch=256), 256 * 200)
bad += soak("one context, batch 64", lambda s: ctx.run_sequence(s, delta, batch=64), 256 * 200)
# every launch carries a different set of per-pose records (period 251 against launches of 256 / 32 poses): a record
# array entry that a kernel read stale (scalar cache, ring reuse) would show up as another pose's block
ref1 = ctx.run_sequence(poses, delta, batch=256)
for b in (256, 32):
    idx = np.arange(b * 400) % 251
    t0 = time.perf_counter(); n_bad = 0; total = 0
    while time.perf_counter() - t0 < seconds / 2:
        out = ctx.run_sequence(poses[idx], delta, batch=b)
        n_bad += int((bits(out) != bits(ref1[idx])).any(axis=1).sum()); total += len(idx)
        idx = (idx + 17) % 251
    print(f"rotating records, batch {b:3d}    {total:9d} evaluations, mismatching blocks: {n_bad}")
    bad += n_bad
m = capi.multi_from_pair(pair, 8, devices=[0, 0, 0], partition=capi.PARTITION_INTERLEAVED)
m.compute_href(pair.pose_init)
bad += soak("3 interleaved shards, host sum", lambda s: m.run_sequence(s, delta, batch=256, group=2), 256 * 100)
r = capi.multi_from_pair(pair, 8, devices=[0], rank=0, world=1)
r.compute_href(pair.pose_init)
r.comm_init(capi.rccl_unique_id())
bad += soak("RCCL, one rank", lambda s: r.run_sequence(s, delta, batch=256, group=2), 256 * 100)
sys.exit(1 if bad else 0)
