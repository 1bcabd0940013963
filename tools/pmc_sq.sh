#!/bin/bash
# tools/pmc_sq.sh <sf> <F> <lib> [<lib> ...] : the two SQ counter passes of tools/pmc.sh for several builds under build_ab/
# (A/B of kernel experiments); summaries under gpurun_out/pmc_<lib>.txt
sf=$1; f=$2; shift 2
R=$PWD; export TMPDIR=/tmp; cd /tmp
for n in "$@"; do
  export LORA_HIP_LIB=$R/build_ab/$n.so
  rm -rf $R/gpurun_out/pmc_$n; mkdir -p $R/gpurun_out/pmc_$n
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_$n/p$i -- python3 $R/tools/prof_one.py $sf $f 2 > $R/gpurun_out/pmc_$n/p$i.log 2>&1 || { echo "pass $i of $n failed"; tail -3 $R/gpurun_out/pmc_$n/p$i.log; exit 1; }
  done
  echo "== $n"; python3 $R/tools/pmc_read.py $R/gpurun_out/pmc_$n | tee $R/gpurun_out/pmc_$n.txt
done
