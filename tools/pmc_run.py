"""Workload for counter passes: N batched evaluation launches, one at a time, on one stream.
Usage (on the GPU box): rocprofv3 --pmc C1 C2 .. -d gpurun_out/pmcX -- python3 tools/pmc_run.py [A|B] [bins] [launches] [poses per launch] [flash]
(flash: the pair with the saturating hot spot, black / saturated patches and depth holes -- round 6)
(round 5: 256 poses per launch by default -- the bench's launch size; 16-pose launches are half tail)"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
pair = synth.make_pair(cfg, **(dict(flash=True, edge_cases=True) if len(sys.argv) > 5 and sys.argv[5] == 'flash' else {}))
ctx = capi.from_pair(pair, bins)
ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
ppl = int(sys.argv[4]) if len(sys.argv) > 4 else 256
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-5 * k, 0, 0], [0, 1e-5 * k, 0]) for k in range(ppl)])
for _ in range(n):
    ctx.launch_batch(0, poses, delta)
    for k in range(ppl):
        ctx.wait(k)
