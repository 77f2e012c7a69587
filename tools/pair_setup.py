#!/usr/bin/env python3
"""tools/pair_setup.py [A|B] [bins]: where a frame pair's time goes on the host stack (nid_host_run_lm = the reference
driver's sequence: Calculate3Dpoint, CudaComputeHref, graph, optimize(10)): the steps' microseconds as the libraries
stamp them under NID_LEGACY_TRACE=1, third of three runs (library and clocks warm) -- in the three verification modes of
the legacy operators (include/nid/legacy_ops.h): the default (background verification), every call (the "per-call
verification" line is what that costs per call, traced on a pair's second call), trusted buffers (round 4's behaviour)."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "A"
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if os.environ.get("NID_PAIR_SETUP_CHILD"):
    sys.path.insert(0, ROOT)
    import time
    synth = importlib.import_module("nid-pose-estimation_amd.synth")
    hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
    pair = synth.make_pair(cfg)
    for fused in (2, 2, 2, 0):
        print(f"[run] fused={fused}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        hostlib.run_lm(pair, bins, pair.pose_init, 10, fused=fused)
        print(f"[run] wall {1e3 * (time.perf_counter() - t0):.3f} ms, optimize() {1e3 * hostlib.last_optimize_seconds():.3f} ms", file=sys.stderr, flush=True)
    sys.exit(0)
for mode, extra in (("DEFAULT, NID_LEGACY_VERIFY_BACKGROUND: cheap keys per call, full hashes by the pool's workers beside the caller", {}),
                    ("NID_LEGACY_VERIFY_EVERY_CALL: every CudaComputeH call verifies the caller's buffers by full hash before it evaluates", {"NID_LEGACY_VERIFY_EVERY_CALL": "1"}),
                    ("NID_LEGACY_VERIFY_TRUSTED (nid_legacy_set_trust_buffers(1) / NID_LEGACY_TRUST_BUFFERS=1: round 4's behaviour)", {"NID_LEGACY_TRUST_BUFFERS": "1"})):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), cfg, str(bins)], capture_output=True, text=True,
                       env=dict(os.environ, NID_PAIR_SETUP_CHILD="1", NID_LEGACY_TRACE="1", **extra))
    runs = p.stderr.split("[run] fused=")
    print(f"=== {mode}")
    print(f"config {cfg}, {bins} bins: steps of one frame pair on the host stack (us); runs 3 (fused + batched trials) and 4 (reference flow) of 4")
    for r in runs[3:]:
        print("--- fused =", r.strip().splitlines()[0])
        for l in r.strip().splitlines()[1:]:
            if l.startswith("[nid trace]") or l.startswith("[run]"):
                print("   ", l.replace("[nid trace] ", ""))
    if p.returncode:
        print(p.stderr[-2000:])
