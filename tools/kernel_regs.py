#!/usr/bin/env python3
"""tools/kernel_regs.py [--keep FILE.s] [-DFLAG=...]: compile the DEVICE side of nid_capi.hip for gfx950 to assembly and
list VGPR / SGPR / spill / LDS / scratch and code bytes of every kernel (no GPU needed).  --keep leaves the assembly in
FILE.s (tools/loop_census.py and tools/isa_rounds.py read it)."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
keep = None
if "--keep" in args:
    i = args.index("--keep")
    keep = args[i + 1]
    del args[i:i + 2]
with tempfile.TemporaryDirectory() as d:
    out = keep or (d + "/dev.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
           *args, "-I", root + "/include", "-I", root + "/nid-pose-estimation_amd/csrc",
           "-o", out, root + "/nid-pose-estimation_amd/csrc/nid_capi.hip"]
    subprocess.check_call(cmd, cwd=d)
    txt = open(out).read()
    sizes = {m.group(1): m.group(0).count("\n") for m in re.finditer(r'^(_Z\S+):\n.*?\n\s+s_endpgm', txt, re.S | re.M)}
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
        body = m.group(2)
        g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [0, "?"])[1]
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} "
              f"scratch {g('private_segment_fixed_size'):>5} lines {sizes.get(m.group(1), 0):>6}  {name[:120]}")
