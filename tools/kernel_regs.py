#!/usr/bin/env python3
"""tools/kernel_regs.py [-DFLAG=...]: compile nid_capi.hip for gfx950 with --save-temps and list VGPR / SGPR /
spill / LDS / scratch of every kernel (no GPU needed)."""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           *sys.argv[1:], "-I", root + "/include", "-I", root + "/nid-pose-estimation_amd/csrc",
           "-o", d + "/lib.so", root + "/nid-pose-estimation_amd/csrc/nid_capi.hip", "--save-temps=obj"]
    subprocess.check_call(cmd, cwd=d)
    txt = open(glob.glob(d + "/*gfx950*.s")[0]).read()
    for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
        body = m.group(2)
        g = lambda k: (re.search(r'\.%s:\s+(\d+)' % k, body) or [0, "?"])[1]
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>4} lds {g('group_segment_fixed_size'):>6} "
              f"scratch {g('private_segment_fixed_size'):>5}  {name[:110]}")
