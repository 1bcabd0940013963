#!/bin/bash
# tools/pmc.sh <tag> <sf> <F> : PMC passes for one kernel config; results under gpurun_out/pmc_<tag>/
tag=$1; sf=$2; f=$3
R=$PWD; export TMPDIR=/tmp; cd /tmp; mkdir -p $R/gpurun_out/pmc_$tag
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 $R/tools/prof_one.py $sf $f 2 > $R/gpurun_out/pmc_$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_$tag/p$i.log; }
done
