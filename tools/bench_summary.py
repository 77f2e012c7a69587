#!/usr/bin/env python3
"""tools/bench_summary.py FILE: the figures of a bench.py line that DESIGN.md quotes."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"]), "it/s; sustained", round(r.get("sustained", {}).get("it_per_s", 0)), "kernel_ms", r["kernel_ms"], "frac", round(r["frac"], 4), "valu busy", round(r.get("valu_pipe", {}).get("busy_frac", float("nan")), 4),
      "issue frac", round(r.get("issue_bound", {}).get("frac", 0), 3))
if "sequential" in r:
    s = r["sequential"]
    print("sequential:", s["form"], round(s["us_per_evaluation"], 2), "us;",
          {k: round(v.get("us_per_evaluation", float("nan")), 2) for k, v in (("launched DIRECT", s["latency_shape_512"]), ("in-launch", s["latency_shape_512"].get("in_launch_reduction", {"us_per_evaluation": 0})),
                                                         ("resident", s["latency_shape_512"].get("resident_evaluator", {"us_per_evaluation": 0})), ("128 threads", s["throughput_shape_128"]))})
    print("  resident stats", s["latency_shape_512"].get("resident_evaluator", {}).get("stats"))
for k in ("cold", "other_math_mode", "flash_pair"):
    if k in r:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in r[k].items() if a != "note"})
if "pose_error_vs_ref" in d:
    p = d["pose_error_vs_ref"]
    print("optimize_only_s", {k: round(v * 1e3, 3) for k, v in p["optimize_only_s"].items()}, "(ms)")
    print("lm outer it/s", round(p["lm_outer_iterations_per_s"]), "same bits across fused flows", p.get("all_flows_same_pose_bits"), "pose diff vs oracle", p["max_abs_minimal_vector_diff"],
          "same trace", p["same_lm_trace"])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu baseline", round(c["value"], 2), "it/s on", c["cores"], "core;", round(c["all_cores"]["value"], 1), "on", c["all_cores"]["nproc"])
