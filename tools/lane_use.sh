#!/bin/bash
# tools/lane_use.sh TAG : VALU lane utilisation of the hot kernel on the plain and on the flash pair (config A, 8 bins, 256-pose launches):
# one counter pass each of SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (counters only).  Table -> gpurun_out/TAG_lanes/table_*.txt
tag=$1
R=$(pwd)
O=$R/gpurun_out/${tag}_lanes
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for data in plain flash; do
  arg=""; [ $data = flash ] && arg=flash
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/$data -- python3 $R/tools/pmc_run.py A 8 8 256 $arg > $O/$data.log 2>&1
  (cd $R && python3 tools/pmc_table.py $O/$data | tee $O/table_$data.txt)
done
