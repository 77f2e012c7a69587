#!/usr/bin/env python3
"""tools/build_variant.py NAME [--src DIR] [-DFLAG=... | other compiler flags]: an experiment build of the HIP library
into exp/libnid_NAME.so (load it with NID_HIP_LIB=exp/libnid_NAME.so; gate it with tests/variant_check.py).  --src: the
root of ANOTHER source tree (e.g. a `git worktree` of an earlier commit: same-box A/B runs of two kernels); trees from
before round 4 are one translation unit.  The translation units are compiled in parallel.  --units A,B: the extra flags go to
the translation units whose file name contains A or B only (e.g. --units nt128,nt256 -mllvm -amdgpu-sched-strategy=max-ilp).  The product's own per-unit flags
(csrc/UNIT_FLAGS) are applied first; NID_NO_UNIT_FLAGS=1 leaves them out."""
import glob, os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
name = args.pop(0)
src = root
if "--src" in args:
    i = args.index("--src"); src = os.path.abspath(args[i + 1]); del args[i:i + 2]
only_units = None
if "--units" in args:
    i = args.index("--units"); only_units = args[i + 1].split(","); del args[i:i + 2]
csrc = os.path.join(src, "nid-pose-estimation_amd", "csrc")
units = sorted(glob.glob(os.path.join(csrc, "*.hip"))) + [os.path.join(csrc, "nid_hostsum.cpp")]
units = [u for u in units if os.path.exists(u)]
os.makedirs(os.path.join(root, "exp"), exist_ok=True)
with tempfile.TemporaryDirectory() as d:
    def unit_flags(u):  # the product's own per-unit flags (csrc/UNIT_FLAGS of the tree that is built; older trees have none)
        out, path = [], os.path.join(csrc, "UNIT_FLAGS")
        for line in (open(path) if os.path.exists(path) and not os.environ.get("NID_NO_UNIT_FLAGS") else []):
            line = line.split("#", 1)[0].strip()
            if line and line.split(":", 1)[0].strip() in os.path.basename(u):
                out += line.split(":", 1)[1].split()
        return out
    def flags(u):
        extra = args if only_units is None or any(k in os.path.basename(u) for k in only_units) else []
        return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", *unit_flags(u), *extra, "-I", os.path.join(src, "include"), "-I", csrc]
    jobs = [["/opt/rocm/bin/hipcc"] + flags(u) + ["-c", "-o", os.path.join(d, os.path.basename(u) + ".o"), u] for u in units]
    with ThreadPoolExecutor(min(8, len(jobs))) as pool:
        list(pool.map(subprocess.check_call, jobs))
    out = os.path.join(root, "exp", f"libnid_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [j[-2] for j in jobs] + ["-ldl"])
    print("built", out, "from", src, "flags", args)
