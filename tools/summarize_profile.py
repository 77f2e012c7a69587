#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small, tracked files under profiles/.
Usage: tools/summarize_profile.py TAG TRACE_DIR [PMC_FETCH_DIR PMC_WRITE_DIR]
Writes profiles/TAG_kernel_stats.csv (verbatim rocprofv3 --stats table),
profiles/TAG_summary.json (per-kernel average duration, counters per dispatch, corrected HBM traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return hits[0] if hits else None


def counter_avg(d, counter):
    f = find(d, "*counter_collection.csv")
    if not f:
        return {}
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = find(trace, "*kernel_stats.csv")
    summary = {"tag": tag, "kernels": {}}
    if stats:
        shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
        with open(stats) as fh:
            for row in csv.DictReader(fh):
                summary["kernels"][row["Name"]] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                                   "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"]),
                                                   "pct": float(row["Percentage"])}
    if len(sys.argv) >= 5:
        fetch = counter_avg(sys.argv[3], "FETCH_SIZE")
        write = counter_avg(sys.argv[4], "WRITE_SIZE")
        for k in set(fetch) | set(write):
            e = summary["kernels"].setdefault(k, {})
            f_kb, w_kb = fetch.get(k), write.get(k)
            e["FETCH_SIZE_KB_per_dispatch"] = f_kb
            e["WRITE_SIZE_KB_per_dispatch"] = w_kb
            if f_kb is not None and w_kb is not None:
                # MI355X_MICROARCH.md section HBM: counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the
                # bytes of a wide coalesced read -> double it (8-B/lane loads are uncalibrated: upper bound)
                e["hbm_bytes_per_dispatch_corrected"] = (2.0 * f_kb + w_kb) * 1024.0
                e["hbm_bytes_per_dispatch_raw"] = (f_kb + w_kb) * 1024.0
    with open(os.path.join(out, f"{tag}_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
