#!/usr/bin/env python3
# ARCHIVED: one-off diagnostic of a parity case; kept because profiles/ and HISTORY.md cite its output (as tools/diag_lm_trace.py). Not part of the test or measurement flow.
"""tools/diag_lm_trace.py [bins]: per-iteration LM trace (trials, chi2, lambda, rho) of the host stack vs the oracle on
the flash + edge-case pair."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
synth = importlib.import_module("nid-pose-estimation_amd.synth")
from oracle import oracle_py as O
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pair = synth.make_pair("A", flash=True, edge_cases=True)
o = O.from_pair(pair, nb, jac_bound="cpu", xform="matrix")
o.compute_href(pair.pose_init)
pose_o, recs_o = o.lm(pair.pose_init, 10)
def show(tag, recs):
    for r in recs:
        print(f"  {tag} it {r['iteration']}: trials {r['lm_trials']}  chi2 {r['chi2']:.12f}  lambda {r['lambda_']:.9e}  rho {r['rho']:.6e}")
show("oracle", recs_o)
for strict in (False, True):
    for fused in (0, 1, 2):
        pose, recs, log = hostlib.run_lm(pair, nb, pair.pose_init, 10, strict=strict, fused=fused)
        print(f"--- {'STRICT' if strict else 'FAST'} fused={fused}")
        show("hip   ", recs)
