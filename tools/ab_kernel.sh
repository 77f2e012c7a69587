#!/bin/bash
# tools/ab_kernel.sh OUT ROUNDS LIB...: same-box A/B of builds of libnid_hip.so (NID_HIP_LIB): kernel_ms (median per-launch
# duration of 256-pose launches, HIP events) and the pipelined rate of bench.py --quick, the libraries taking turns, ROUNDS times each.
out=$1; n=$2; shift 2
mkdir -p "$(dirname $out)"; : > $out
for i in $(seq $n); do
  for lib in "$@"; do
    NID_HIP_LIB=$lib python bench.py --quick --no-cpu-baseline --steps 200000 --warmup 20000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'kernel_ms %.4f' % r['kernel_ms'], 'value %.0f' % d['value'])" >> $out
  done
done
cat $out
