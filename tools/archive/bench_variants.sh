#!/bin/bash
# ARCHIVED: round 2-3 A/B driver over build variants (superseded by tools/ab_kernel.sh); output recorded in / cited by: profiles/README.md (there as tools/bench_variants.sh). Not part of the test or measurement flow.
# tools/bench_variants.sh OUTDIR NAME...: bench.py (N=1, no CPU legs) once per experiment build exp/libnid_NAME.so
# (tools/build_variant.py), one line per variant: sustained it/s, kernel ms for 64 poses alone.
out=$1; shift
mkdir -p $out
for v in "$@"; do
  NID_HIP_LIB=exp/libnid_$v.so python bench.py --no-cpu-baseline --quick --steps 200000 --warmup 20000 > $out/bench_$v.json 2> $out/bench_$v.err
  python - "$v" "$out/bench_$v.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:10s} {d['value']:10.0f} it/s  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us  frac {d['roofline']['frac']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
