#!/bin/bash
# tools/profile_round.sh TAG   (run on the GPU box, from the repo root, e.g. through gpurun)
# For configs A and B: rocprofv3 kernel trace + stats of bench.py with ONE launch in flight (the duration
# bench.py's roofline block measures with HIP events), the same with the default two-stream pipeline,
# and the FETCH_SIZE / WRITE_SIZE counter passes (separate runs, as the HBM section of
# MI355X_MICROARCH.md prescribes).  Raw output -> gpurun_out/TAG_<cfg>/; condense with
# tools/summarize_profile.py afterwards (on any machine).  The two trace passes run the whole bench (its sustained leg
# keeps the clocks up: with --quick the few launches between host work average 3 % longer; --no-flash: the flash-pair leg
# launches the SAME kernel on other data and would be averaged into its line); the counter passes are --quick.
tag=$1
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cfg in A B; do
  steps=4000; [ $cfg = B ] && steps=1600
  O=$R/gpurun_out/${tag}_$cfg
  rm -rf $O
  export NID_ONE_STREAM=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --config $cfg --steps $steps --warmup 200 --no-cpu-baseline --no-flash > $R/gpurun_out/${tag}_${cfg}_bench_onestream.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --config $cfg --steps 400 --warmup 40 --no-cpu-baseline --quick > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --config $cfg --steps 400 --warmup 40 --no-cpu-baseline --quick > /dev/null 2>&1
  unset NID_ONE_STREAM
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pipelined -- python3 $R/bench.py --config $cfg --steps $steps --warmup 200 --no-cpu-baseline --no-flash > $R/gpurun_out/${tag}_${cfg}_bench_pipelined.json 2>/dev/null
done
cd $R
find gpurun_out/${tag}_A gpurun_out/${tag}_B -name "*kernel_stats.csv" | head
