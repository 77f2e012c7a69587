"""tools/lm_trace.py [fused ...] [resident]: NID_LM_TRACE stages of every outer iteration of the host LM (stderr)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NID_LM_TRACE"] = "1"
synth = importlib.import_module("nid-pose-estimation_amd.synth")
hostlib = importlib.import_module("nid-pose-estimation_amd.hostlib")
pair = synth.make_pair("A")
args = [a for a in sys.argv[1:]]
resident = "resident" in args
flows = [int(a) for a in args if a.isdigit()] or [2, 3]
hostlib.set_resident(resident)
for fused in flows:
    for rep in range(3):
        print("--- run", rep, "fused", fused, "resident", resident, file=sys.stderr)
        pose, recs, _ = hostlib.run_lm(pair, 8, pair.pose_init, 10, fused=fused)
        print("optimize s", hostlib.last_optimize_seconds(), len(recs), [r['lm_trials'] for r in recs], file=sys.stderr)
hostlib.set_resident(False)
