#!/bin/bash
# tools/final_round.sh TAG: the measurements DESIGN.md section 7 quotes, in one go on the GPU box (through gpurun):
# rocprofv3 kernel stats + HBM counter passes (tools/profile_round.sh), SQ counters (tools/pmc_round.sh), the bench lines
# (default flags, config B, driver-style flags) and the latency sweeps.  Output under gpurun_out/.
# PART (second argument): 1 = bench lines, kernel stats, counter passes; 2 = everything else; empty = both (> 20 minutes: two gpurun calls)
tag=$1; part=$2
mkdir -p gpurun_out/$tag
if [ -z "$part" ] || [ "$part" = 1 ]; then
python bench.py > gpurun_out/$tag/bench_A.json 2> gpurun_out/$tag/bench_A.err; echo "bench A rc=$?"
bash tools/profile_round.sh $tag > gpurun_out/$tag/profile_round.log 2>&1; echo "profile rc=$?"
bash tools/pmc_round.sh $tag A > gpurun_out/$tag/pmc_round.log 2>&1; echo "pmc rc=$?"
bash tools/pmc_round.sh ${tag}B B > gpurun_out/$tag/pmc_round_B.log 2>&1; echo "pmc B rc=$?"
fi
if [ -z "$part" ] || [ "$part" = 2 ]; then
python bench.py --config B > gpurun_out/$tag/bench_B.json 2> gpurun_out/$tag/bench_B.err; echo "bench B rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/$tag/bench_A_driver_flags.json 2> gpurun_out/$tag/bench_A_driver_flags.err; echo "bench driver rc=$?"
python tools/latency_sweep.py A 8 > gpurun_out/$tag/launch_cost_A.txt 2> gpurun_out/$tag/launch_cost_A.err; echo "lat A rc=$?"
python tools/latency_sweep.py B 8 > gpurun_out/$tag/launch_cost_B.txt 2> gpurun_out/$tag/launch_cost_B.err; echo "lat B rc=$?"
python tools/batch_sweep.py > gpurun_out/$tag/batch_sweep.txt 2> gpurun_out/$tag/batch_sweep.err; echo "batch rc=$?"
python tools/direct_latency.py A 8 > gpurun_out/$tag/direct_latency_A.txt 2> gpurun_out/$tag/direct_latency_A.err; echo "direct A rc=$?"
python tools/direct_latency.py B 8 > gpurun_out/$tag/direct_latency_B.txt 2> gpurun_out/$tag/direct_latency_B.err; echo "direct B rc=$?"
python tools/lm_latency.py A 8 > gpurun_out/$tag/lm_latency_A.txt 2> gpurun_out/$tag/lm_latency_A.err; echo "lm rc=$?"
python tools/lm_trace.py 4 resident > /dev/null 2> gpurun_out/$tag/lm_trace_fused4_resident.txt; echo "lm trace rc=$?"
python tools/flash_rate.py > gpurun_out/$tag/flash_rate.txt 2> gpurun_out/$tag/flash_rate.err; echo "flash rc=$?"
./tools/ubench/mailbox_latency > gpurun_out/$tag/mailbox_latency.txt 2>&1; echo "mailbox rc=$?"
NID_DIRECT_TRACE=1 python tools/direct_trace.py > gpurun_out/$tag/direct_trace.txt 2>&1; echo "direct trace rc=$?"
python tools/short_seq_sweep.py A 8 > gpurun_out/$tag/short_seq_A.txt 2> gpurun_out/$tag/short_seq_A.err; echo "short seq rc=$?"
python tools/timed_region_probe.py > gpurun_out/$tag/timed_region_probe.txt 2> gpurun_out/$tag/timed_region_probe.err; echo "probe rc=$?"
python tools/pair_setup.py A 8 > gpurun_out/$tag/pair_setup.txt 2> gpurun_out/$tag/pair_setup.err; echo "pair setup rc=$?"
python tools/legacy_call_cost.py A 8 > gpurun_out/$tag/legacy_call_cost.txt 2> gpurun_out/$tag/legacy_call_cost.err; echo "legacy call cost rc=$?"
./tools/ubench/div_shared > gpurun_out/$tag/div_shared.txt 2>&1; echo "div_shared rc=$?"
[ -f exp/libnid_hip_r6base.so ] && ROUNDS=2 python tools/flash_ab.py exp/libnid_hip_r6base.so default > gpurun_out/$tag/flash_ab.txt 2>&1; echo "flash ab rc=$?"
( R=$(pwd); cd /tmp && export TMPDIR=/tmp && NID_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_A10/trace -- python3 $R/bench.py --bins 10 --steps 4000 --warmup 200 --no-cpu-baseline --no-flash > $R/gpurun_out/${tag}_A10_bench_onestream.json 2>/dev/null ); echo "bins10 trace rc=$?"
bash tools/pmc_round.sh ${tag}A10 A 256 10 > gpurun_out/$tag/pmc_round_A10.log 2>&1; echo "pmc A10 rc=$?"
( R=$(pwd); cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_flash/trace -- python3 $R/tools/flash_profile.py flash FAST > $R/gpurun_out/$tag/flash_profile.txt 2>&1 ); echo "flash profile rc=$?"
./tools/ubench/valu_wallclock > gpurun_out/$tag/valu_wallclock.txt 2>&1; echo "valu wallclock rc=$?"
python tools/first_shot_probe.py > gpurun_out/$tag/first_shot_probe.txt 2> gpurun_out/$tag/first_shot_probe.err; echo "first shot probe rc=$?"
fi
