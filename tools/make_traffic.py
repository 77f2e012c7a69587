#!/usr/bin/env python3
"""profiles/traffic.json from profiles/TAG_{A,B}_summary.json (tools/summarize_profile.py output):
HBM bytes per launch of the hot kernel = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md, HBM
section: counters in KB, gfx950 FETCH_SIZE reports half of a wide coalesced read).
Usage: tools/make_traffic.py TAG POSES_PER_LAUNCH [BINS]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ppl = sys.argv[1], int(sys.argv[2])
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 8
out = {}
for c in "AB":
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_{c}_summary.json")))
    for k, e in d["kernels"].items():
        # the launches of the timed region: > 16 poses -> the EXT = true instantiation
        if re.search(rf"k_eval2<\d+, true, false, {bins}, false, {'true' if ppl > 16 else 'false'}(, \d+)?(, (true|false))?>", k) \
                and "hbm_bytes_per_dispatch_corrected" in e:
            out[f"{c}:{bins}"] = {
                "kernel": k, "poses_per_launch": ppl,
                "hbm_bytes_per_launch": e["hbm_bytes_per_dispatch_corrected"],
                "hbm_bytes_per_pose": e["hbm_bytes_per_dispatch_corrected"] / ppl,
                "raw_bytes_per_launch": e["hbm_bytes_per_dispatch_raw"],
                "avg_ns_under_rocprof": e.get("avg_ns"),
                "source": f"profiles/{tag}_{c}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; "
                          "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section; 8-B/lane loads are "
                          "uncalibrated, so this is an upper bound)"}
            print(c, e.get("avg_ns"), e["hbm_bytes_per_dispatch_corrected"])
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
