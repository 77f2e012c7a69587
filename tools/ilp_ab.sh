mkdir -p gpurun_out/r06
O=gpurun_out/r06/ilp_ab.txt
: > $O
echo "== ten bins" >> $O
NID_AB_BINS=10 ROUNDS=2 python tools/flash_ab.py default exp/libnid_ilp.so >> $O 2>&1
for lib in default ilp default ilp; do
  echo "== dependent evaluations, $lib" >> $O
  if [ $lib = default ]; then python tools/direct_latency.py A 8 >> $O 2>&1; else NID_HIP_LIB=$PWD/exp/libnid_ilp.so python tools/direct_latency.py A 8 >> $O 2>&1; fi
done
for lib in default ilp; do
  echo "== flash_rate (FAST / STRICT), $lib" >> $O
  if [ $lib = default ]; then python tools/flash_rate.py >> $O 2>&1; else NID_HIP_LIB=$PWD/exp/libnid_ilp.so python tools/flash_rate.py >> $O 2>&1; fi
done
for lib in default ilp default ilp; do
  echo "== bench config B, $lib" >> $O
  if [ $lib = default ]; then python bench.py --config B --quick --no-cpu-baseline --no-flash 2>/dev/null | python tools/bench_summary.py /dev/stdin 2>/dev/null | head -3 >> $O; else NID_HIP_LIB=$PWD/exp/libnid_ilp.so python bench.py --config B --quick --no-cpu-baseline --no-flash 2>/dev/null | python tools/bench_summary.py /dev/stdin 2>/dev/null | head -3 >> $O; fi
done
cat $O
