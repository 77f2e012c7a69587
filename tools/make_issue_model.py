#!/usr/bin/env python3
"""profiles/issue_model.json from a tools/pmc_round.sh table (gpurun_out/TAG_pmc/table.txt): instructions per wave of
the evaluation kernel by class, from the SQ_INSTS_* counters of 16-pose launches (tools/pmc_run.py).  bench.py turns it
into roofline.issue_bound = wave-instructions per second / the SIMDs' issue peak.
Usage: tools/make_issue_model.py TABLE CONFIG BINS POSES_PER_LAUNCH   (e.g. gpurun_out/r02_pmc/table.txt A 8 16)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table, cfg, bins, ppl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
c = {}
for line in open(table):
    p = line.split()
    if len(p) >= 2:
        try:
            c[p[0]] = float(p[1])
        except ValueError:
            pass
w = c["SQ_WAVES"]
path = os.path.join(ROOT, "profiles", "issue_model.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out[f"{cfg}:{bins}"] = {
    "valu": c["SQ_INSTS_VALU"] / w, "salu": c["SQ_INSTS_SALU"] / w, "lds": c["SQ_INSTS_LDS"] / w,
    "vmem": (c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]) / w, "smem": c["SQ_INSTS_SMEM"] / w,
    "branch": c["SQ_INSTS_BRANCH"] / w, "waves_per_pose": w / ppl,
    "wave_cycles": c["SQ_WAVE_CYCLES"] / w, "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], "active_inst_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
    # the vector pipe: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs, GRBM_GUI_ACTIVE cycles summed over the 8
    # XCDs (MI355X_MICROARCH.md), both per launch: busy = 4 x active / (cycles of the launch x 1024 SIMDs)
    "poses_per_launch": ppl,
    "valu_cycles_per_instruction": 4.0 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"],
    "valu_busy_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0),
    "lds_busy_frac": 4.0 * c["SQ_ACTIVE_INST_LDS"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if "SQ_ACTIVE_INST_LDS" in c else None,
    "launch_cycles": c["GRBM_GUI_ACTIVE"] / 8.0,
    "source": f"{os.path.relpath(table, ROOT)} (rocprofv3 --pmc SQ_* passes over tools/pmc_run.py, {ppl} poses per launch, "
              "128-thread workgroups, FAST math; committed as profiles/r0N_A_pmc_counters.txt of the same round)"}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out[f"{cfg}:{bins}"], indent=1))
