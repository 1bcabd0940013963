#!/bin/bash
# timing-only ablation builds (results are wrong by construction): tools/abl.sh "12,2" abl33 abl37 ...
cfgs=$1; shift
for n in "$@"; do
  for b in ${BLOCKS:-0}; do
    echo "== $n blocks/CU=$b"
    if [ "$b" = 0 ]; then unset LORA_HIP_BLOCKS_PER_CU; else export LORA_HIP_BLOCKS_PER_CU=$b; fi
    LORA_HIP_LIB=$PWD/build_ab/$n.so timeout -k 10 150 python -u tools/kbench.py $cfgs 2>&1 | grep "SF"
  done
done
exit 0
