#!/usr/bin/env python3
"""tools/kernel_regs.py [--keep DIR] [--only SUBSTR] [-DFLAG=...]: compile the DEVICE side of every translation unit of
csrc/ for gfx950 to assembly (in parallel) and list VGPR / SGPR / spill / scratch of every kernel (no GPU needed).
--keep leaves the assembly files in DIR (tools/loop_census.py reads them); --only: translation units whose file name
contains SUBSTR."""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
keep = only = None
for flag in ("--keep", "--only"):
    if flag in args:
        i = args.index(flag)
        if flag == "--keep": keep = args[i + 1]
        else: only = args[i + 1]
        del args[i:i + 2]
csrc = root + "/nid-pose-estimation_amd/csrc"
units = [u for u in sorted(glob.glob(csrc + "/*.hip")) if not only or only in os.path.basename(u)]
with tempfile.TemporaryDirectory() as d:
    outdir = keep or d
    os.makedirs(outdir, exist_ok=True)
    def build(u):
        out = os.path.join(outdir, os.path.basename(u)[:-4] + ".s")
        unit = []  # the product's own per-unit flags (csrc/UNIT_FLAGS)
        for line in (open(csrc + "/UNIT_FLAGS") if os.path.exists(csrc + "/UNIT_FLAGS") and not os.environ.get("NID_NO_UNIT_FLAGS") else []):
            line = line.split("#", 1)[0].strip()
            if line and line.split(":", 1)[0].strip() in os.path.basename(u):
                unit += line.split(":", 1)[1].split()
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                               *unit, *args, "-I", root + "/include", "-I", csrc, "-o", out, u], stderr=subprocess.DEVNULL)
        return out
    with ThreadPoolExecutor(8) as pool:
        outs = list(pool.map(build, units))
    for out in outs:
        txt = open(out).read()
        for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
            body = m.group(2)
            g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [0, "?"])[1]
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5}  {name[:120]}")
