#!/bin/bash
# tools/pmc_round.sh TAG [A|B] [POSES] [BINS] [flash] : SQ / TA / TCP / TCC counter passes over tools/pmc_run.py (POSES-pose launches of the hot
# kernel -- default 256, the bench's launch size --, one at a time), a few counters per pass (counters only: no trace domains next to --pmc).
# Raw output -> gpurun_out/TAG_pmc/passN; table -> gpurun_out/TAG_pmc/table.txt (tools/pmc_table.py).
tag=$1; cfg=${2:-A}; ppl=${3:-256}; bins=${4:-8}; data=${5:-}
R=$(pwd)
O=$R/gpurun_out/${tag}_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" "TCP_TOTAL_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pass$i -- python3 $R/tools/pmc_run.py $cfg $bins 8 $ppl $data > $O/pass$i.log 2>&1
done
cd $R
python3 tools/pmc_table.py $O/pass* | tee $O/table.txt
