#!/bin/bash
# developer tool: the planned block's time against the stretch length (LORA_CORE_STRETCH), one GPU call
#   bash tools/stretch_sweep.sh > gpurun_out/stretch_sweep.txt
for s in 600000 270000 180000 135000 110000 90000 66000; do
  echo "== LORA_CORE_STRETCH=$s"
  LORA_CORE_STRETCH=$s LORA_CORE_TIMING=1 timeout -k 10 120 python tools/pipeline_bench.py 2>&1 | grep -B3 "device-resident" | grep -v "first pass\|SFD stage" || exit 1
done
