import importlib, sys, numpy as np
sys.path.insert(0, '.')
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
pair = synth.make_pair("B")
ctx = capi.from_pair(pair, 8)
ctx.compute_href(pair.pose_init)
ctx.set_launch_shape(512, 512)
try:
    ctx.set_resident(True)
    print("set_resident ok")
    print(ctx.run_chain(np.stack([pair.pose_init] * 10), float(np.sqrt(0.95)), want_jac=True, collect=False)[1])
    print(ctx.resident_stats())
except Exception as e:
    print("first error:", e)
