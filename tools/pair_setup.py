#!/usr/bin/env python3
"""tools/pair_setup.py [A|B] [bins]: where a frame pair's time goes on the host stack (nid_host_run_lm = the reference
driver's sequence: Calculate3Dpoint, CudaComputeHref, graph, optimize(10)): the steps' microseconds as the libraries
stamp them under NID_LEGACY_TRACE=1, third of three runs (library and clocks warm) -- in the verification modes of
the legacy operators (include/nid/legacy_ops.h): the default (rotating: one slice of 16 per call, inside the call), the same
without pool threads, every slice on every call, trusted buffers.  The per-call cost of a mode is its optimize() of the
reference flow (fused = 0: one CudaComputeH per evaluation) against the trusted mode's."""
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
for mode, extra in (("DEFAULT, NID_LEGACY_VERIFY_ROTATING: cheap keys + 3 (cost-only) / 6 (with the Jacobian) of the 128 slices of each big buffer per call, hashed by the pool's workers (polling for 150 us behind a job) WHILE the device evaluates, joined before the call returns; fused flows: NATIVE pair setup (nid_legacy_set_pair_u16)", {}),
                    ("fused flows through the legacy operators' setup (NID_HOST_LEGACY_SETUP=1: round 5's route)", {"NID_HOST_LEGACY_SETUP": "1"}),
                    ("ROTATING, 2 of 128 slices per call (NID_LEGACY_VERIFY_SLICES=2: a change is found within 64 calls)", {"NID_LEGACY_VERIFY_SLICES": "2", "NID_HOST_LEGACY_SETUP": "1"}),
                    ("ROTATING, workers park at once (NID_LEGACY_HASH_SPIN_US=0)", {"NID_LEGACY_HASH_SPIN_US": "0", "NID_HOST_LEGACY_SETUP": "1"}),
                    ("ROTATING without pool threads (NID_LEGACY_HASH_THREADS=0: the caller hashes its slice behind the evaluation)", {"NID_LEGACY_HASH_THREADS": "0", "NID_HOST_LEGACY_SETUP": "1"}),
                    ("NID_LEGACY_VERIFY_EVERY_CALL: all 128 slices on every CudaComputeH call", {"NID_LEGACY_VERIFY_EVERY_CALL": "1", "NID_HOST_LEGACY_SETUP": "1"}),
                    ("NID_LEGACY_VERIFY_TRUSTED (nid_legacy_set_trust_buffers(1) / NID_LEGACY_TRUST_BUFFERS=1: round 4's behaviour)", {"NID_LEGACY_TRUST_BUFFERS": "1", "NID_HOST_LEGACY_SETUP": "1"})):
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
