#!/bin/bash
# tools/build_variant.sh NAME -DFLAG=... : an experiment build of the HIP library into exp/libnid_NAME.so
# (load it with NID_HIP_LIB=exp/libnid_NAME.so; gate it with tests/variant_check.py); prints VGPR / spill counts of the hot kernel.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/exp /tmp/nidv_$name
cd /tmp/nidv_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -ldl "$@" -I$root/include -I$root/nid-pose-estimation_amd/csrc \
  -o /tmp/nidv_$name/libnid_$name.so $root/nid-pose-estimation_amd/csrc/nid_capi.hip $root/nid-pose-estimation_amd/csrc/nid_hostsum.cpp --save-temps=obj 2>&1 | grep -E "error" -A5 || true
python3 - "$name" <<'PY'
import re, glob, sys
f = glob.glob('/tmp/nidv_%s/*gfx950*.s' % sys.argv[1])
txt = open(f[0]).read()
for m in re.finditer(r'\.name:\s+(\S*k_eval2\S*)\n(.*?)\.wavefront_size', txt, re.S):
    body = m.group(2)
    t = re.search(r'k_eval2ILi128ELb(\d)ELb(\d)ELi(\d+)ELb(\d)', m.group(1))
    if t and t.group(3) == '8' and t.group(2) == '0' and t.group(1) == '1' and t.group(4) == '0':
        print(sys.argv[1], "hot kernel vgpr", re.search(r'\.vgpr_count:\s+(\d+)', body).group(1), "spill",
              re.search(r'\.vgpr_spill_count:\s+(\d+)', body).group(1))
PY
mv /tmp/nidv_$name/libnid_$name.so $root/exp/
rm -rf /tmp/nidv_$name
