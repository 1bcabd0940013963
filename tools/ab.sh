#!/bin/bash
# A/B several builds of the library in one GPU call: tools/ab.sh "12,2 12,1 9,2" lib_a lib_b ...
cfgs=$1; shift
for n in "$@"; do
  echo "== $n"
  LORA_HIP_LIB=$PWD/build_ab/$n.so timeout -k 10 120 python -u tools/quick_bench.py $cfgs 2>&1 | grep -v amdgpu.ids || exit 1
done
