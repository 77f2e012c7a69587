# ARCHIVED: one-off diagnostic of a parity case; output recorded in / cited by: profiles/r05_ablations_A.txt item 5 (there as tools/diag_lat_loop.py). Not part of the test or measurement flow.
"""tools/diag_lat_loop.py [NT] [NB]: where the latency form and the loop form of a launch_batch differ (per pose, per cell)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import importlib
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from test_parity_gpu import _poses, _identity_pose, DELTA
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for name, pair in (("A", synth.make_pair("A")), ("A flash", synth.make_pair("A", flash=True)), ("S", synth.make_pair("S"))):
    ctx = capi.from_pair(pair, nb)
    ctx.set_launch_shape(nt, nt)
    ctx.compute_href(pair.pose_init)
    poses = list(_poses(synth, pair).values()) + [_identity_pose(synth, pair)]
    for rep in range(3):
        got = {}
        for loop in (False, True):
            ctx.set_loop_form(loop)
            ctx.repair_count(reset=True)
            ctx.launch_batch(0, poses, DELTA)
            res = [ctx.wait(k) for k in range(len(poses))]
            single = [ctx.normal_equations(p, DELTA) for p in poses]
            got[loop] = (res, single, ctx.repair_count())
        for k in range(len(poses)):
            a, b = got[False][0][k], got[True][0][k]
            sa, sb = got[False][1][k], got[True][1][k]
            print(name, "rep", rep, "pose", k, "chi2 lat %.12f loop %.12f | single lat %.12f loop %.12f | na %s %s | H equal %s" % (
                a[2], b[2], sa[2], sb[2], a[3], b[3], np.array_equal(a[0], b[0])), "repairs", got[False][2], got[True][2])
