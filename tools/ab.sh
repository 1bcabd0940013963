#!/bin/bash
# A/B several builds of the library in one GPU call: tools/ab.sh "12,2 12,1 9,2" lib_a lib_b ...   (libs under build_ab/)
cfgs=$1; shift
for n in "$@"; do
  echo "== $n"
  LORA_HIP_LIB=$PWD/build_ab/$n.so timeout -k 10 150 python -u tools/kbench.py $cfgs 2>&1 | grep -v amdgpu.ids || exit 1
done
