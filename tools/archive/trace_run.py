# ARCHIVED: round 2 helper for a rocprofv3 trace pass; kept because profiles/ and HISTORY.md cite its output (as tools/trace_run.py). Not part of the test or measurement flow.
"""Workload for a kernel trace: 40 launches of 256 poses (cost + Jacobian, FAST), one at a time on one stream, on the
plain pair and on the flash pair.  Usage: rocprofv3 --kernel-trace --stats -d DIR -- python3 tools/trace_run.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
delta = float(np.sqrt(0.95))
for kw in ({}, dict(flash=True, edge_cases=True)) if len(sys.argv) < 2 else ({},):
    pair = synth.make_pair("A", **kw)
    ctx = capi.from_pair(pair, 8)
    ctx.compute_href(pair.pose_init)
    poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
    ctx.run_sequence(poses[np.arange(256 * 40) % 256], delta, batch=256, collect=False)
    for _ in range(4):
        ctx.time_launches(poses, delta, repeats=10)
    ctx.close()
