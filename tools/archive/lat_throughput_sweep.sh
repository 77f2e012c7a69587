#!/bin/bash
# ARCHIVED: round 3 sweep of the latency form's throughput (profiles/r03_*); output recorded in / cited by: profiles/r02_ablations_A.txt, profiles/README.md (there as tools/lat_throughput_sweep.sh). Not part of the test or measurement flow.
# tools/lat_throughput_sweep.sh [THREADS...]: cost + Jacobian throughput at 16 poses per launch (the launches whose
# per-pose records ride in the kernel arguments) per workgroup shape; honours NID_HIP_LIB (tools/build_variant.py).
mkdir -p gpurun_out/lat_sweep
for nt in ${@:-128 256 512 1024}; do
  for b in 16; do
    python bench.py --no-cpu-baseline --quick --steps 100000 --warmup 10000 --batch $b --block-threads $nt > gpurun_out/lat_sweep/lat_${nt}_$b.json 2> gpurun_out/lat_sweep/lat_${nt}_$b.err
    python - $nt $b gpurun_out/lat_sweep/lat_${nt}_$b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(f"threads {sys.argv[1]:5s} batch {sys.argv[2]:3s} {d['value']:10.0f} it/s  kernel {d['roofline']['kernel_ms']*1e3:7.1f} us  frac {d['roofline']['frac']:.3f}")
PY
  done
done
