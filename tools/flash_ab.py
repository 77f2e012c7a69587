#!/usr/bin/env python3
"""tools/flash_ab.py LIB [LIB ...]: same-box A/B of library builds (tools/build_variant.py): for every library, in
round-robin order and ROUNDS times, the kernel time of a 256-pose cost+Jacobian launch (FAST) on the plain and on the
flash pair and the pipelined rate -- each library in its own process (NID_HIP_LIB).  NID_AB_BINS: bin count (8);
NID_AB_THREADS=J,C: launch shape of the cost+Jacobian and cost-only launches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import importlib, os, sys, time, json
import numpy as np
sys.path.insert(0, %r)
capi = importlib.import_module("nid-pose-estimation_amd.capi")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
delta = float(np.sqrt(0.95)); out = {}
for name, kw in (("plain", {}), ("flash", dict(flash=True, edge_cases=True))):
    pair = synth.make_pair("A", **kw)
    ctx = capi.from_pair(pair, int(os.environ.get("NID_AB_BINS", "8")))
    if os.environ.get("NID_AB_THREADS"):  # workgroup shape of cost+Jacobian / cost-only launches (nid_set_launch_shape)
        ctx.set_launch_shape(*[int(v) for v in os.environ["NID_AB_THREADS"].split(",")])
    ctx.compute_href(pair.pose_init)
    poses = np.stack([synth.perturb_pose7(pair.pose_init, [1e-4 * k, 0, 0], [0, 1e-4 * k, 0]) for k in range(256)])
    seq = poses[np.arange(256 * 120) %% 256]
    ctx.run_sequence(seq[:256 * 40], delta, batch=256, collect=False)
    t0 = time.perf_counter(); ctx.run_sequence(seq, delta, batch=256, collect=False); el = time.perf_counter() - t0
    ms = float(np.median([ctx.time_launches(poses, delta, repeats=10) for _ in range(7)]))
    msc = float(np.median([ctx.time_launches(poses, delta, repeats=10, want_jac=False) for _ in range(5)]))
    ctx.repair_count(reset=True); ctx.run_sequence(poses, delta, batch=256, collect=False)
    out[name] = dict(it_per_s=len(seq) / el, kernel_us=ms * 1e3, cost_only_us=msc * 1e3, repairs=ctx.repair_count())
    ctx.close()
print(json.dumps(out))
''' % ROOT
libs = sys.argv[1:]
rounds = int(os.environ.get("ROUNDS", "2"))
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["NID_HIP_LIB"] = os.path.join(ROOT, lib)
        p = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, env=env)
        try:
            o = json.loads(p.stdout.strip().splitlines()[-1])
            print(f"round {r} {lib:28s} plain {o['plain']['kernel_us']:7.1f} us ({o['plain']['it_per_s']:7.0f} it/s; cost-only {o['plain']['cost_only_us']:6.1f})   "
                  f"flash {o['flash']['kernel_us']:7.1f} us ({o['flash']['it_per_s']:7.0f} it/s; cost-only {o['flash']['cost_only_us']:6.1f}; repairs {o['flash']['repairs']})", flush=True)
        except Exception:
            print(f"round {r} {lib}: FAILED\n{p.stderr[-1500:]}", flush=True)
