#!/usr/bin/env python3
"""tools/direct_trace.py: NID_DIRECT_TRACE=1 python tools/direct_trace.py -- where the time of a dependent evaluation goes
on the host side of a DIRECT launch / resident request (first record's arrival, last wait, end of the host's sums)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair("A"); delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-3 * k, 0, 0], [0, 1e-3 * k, 0]) for k in range(16)])
ctx = capi.from_pair(pair, 8); ctx.compute_href(pair.pose_init)
seq = poses[np.arange(4000) % 16]
ctx.run_sequence(seq, delta, batch=16, collect=False)
for nt in (512, 1024):
    ctx.set_launch_shape(nt, nt)
    for resident in (False, True):
        ctx.set_resident(resident)
        for jac in (True, False):
            print("shape", nt, "resident", resident, "jac", jac, flush=True)
            for _ in range(2):
                t = ctx.run_chain(seq[:2000], delta, want_jac=jac, collect=False)[1]
                print("   chain us", t / 2000 * 1e6, flush=True)
