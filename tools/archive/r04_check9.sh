#!/bin/bash
# ARCHIVED: round 4's sweep / check driver script (its output is under profiles/r04_*); kept because profiles/ and HISTORY.md cite its output (as tools/r04_check9.sh). Not part of the test or measurement flow.
mkdir -p gpurun_out/r04t2
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04t2/gpu_tests.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r04t2/gpu_tests.txt
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04t2/bench_driver.json 2> gpurun_out/r04t2/bench_driver.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04t2/bench_driver.json').read().strip().splitlines()[-1])
print('value', round(d['value']), 'sustained', round(d['roofline']['sustained']['it_per_s']), 'kernel_ms', d['roofline']['kernel_ms'], 'frac', round(d['roofline']['frac'],3), 'contract', round(d['roofline']['contract_frac'],3))
print('lm_wall_s', d['pose_error_vs_ref']['lm_wall_s'])
print('flash', d['roofline']['flash_pair']['it_per_s'], d['roofline']['flash_pair']['kernel_ms'])
PY
