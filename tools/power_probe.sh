#!/bin/bash
# usage: tools/power_probe.sh <label> [env assignments...] -- runs a ~5 s pipelined bench and reports the
# median socket power / sclk that rocm-smi showed while it ran (experiments on the power wall)
label=$1; shift
( for i in $(seq 1 200); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr "\n" " "; echo; sleep 0.2; done > gpurun_out/smi_$label.log ) &
SMI=$!
env "$@" python bench.py --no-cpu-baseline --steps 1600000 --warmup 400 > gpurun_out/long_$label.json 2>/dev/null
kill $SMI 2>/dev/null
python - "$label" <<'PY'
import re, sys, json, statistics
label = sys.argv[1]
pw, ck = [], []
for line in open(f"gpurun_out/smi_{label}.log"):
    m = re.search(r"\((\d+)Mhz\).*Power \(W\): ([\d.]+)", line)
    if m and float(m.group(2)) > 400:
        ck.append(int(m.group(1))); pw.append(float(m.group(2)))
d = json.loads(open(f"gpurun_out/long_{label}.json").read().strip().splitlines()[-1])
print(f"{label:12s} {d['value']:10.0f} it/s   kernel {d['roofline']['kernel_ms']*1e3:6.1f} us   power median {statistics.median(pw) if pw else 0:6.0f} W  max {max(pw) if pw else 0:6.0f}   sclk median {statistics.median(ck) if ck else 0}  (n={len(pw)})")
PY
