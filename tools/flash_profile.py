#!/usr/bin/env python3
"""tools/flash_profile.py [plain|flash] [FAST|STRICT] [bins]: 60 launches of 256 poses (cost + Jacobian) on ONE pair in ONE math mode and
nothing else -- the workload for `rocprofv3 --kernel-trace --stats` when the question is how a flash launch's time splits
between k_eval2 and the k_repair behind it (tools/flash_rate.py mixes pairs and modes under the same kernel names)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
which = sys.argv[1] if len(sys.argv) > 1 else "flash"
mode = capi.MATH_STRICT if (len(sys.argv) > 2 and sys.argv[2] == "STRICT") else capi.MATH_FAST
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pair = synth.make_pair("A", **(dict(flash=True, edge_cases=True) if which == "flash" else {}))
ctx = capi.from_pair(pair, bins, math=mode)
ctx.compute_href(pair.pose_init)
delta = float(np.sqrt(0.95))
poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
ms = [ctx.time_launches(poses, delta, repeats=10) for _ in range(6)]
print(f"{which} {'STRICT' if mode == capi.MATH_STRICT else 'FAST'} {bins} bins: kernel {1e3 * float(np.median(ms)):.1f} us per 256-pose launch (HIP events)")
ctx.close()
