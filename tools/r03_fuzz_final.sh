#!/bin/bash
# The round's last fuzz campaign on the final library: four modes, new seeds; progress goes to stdout, the summary lines to
# gpurun_out/r03_fuzz_final.txt.   gpurun --timeout 1150 -- 'bash tools/r03_fuzz_final.sh'
set -e
export PYTHONUNBUFFERED=1
out=gpurun_out/r03_fuzz_final.txt
: > $out
run() {   # run <seconds> <seed> [any]   (environment from the caller)
  timeout -k 10 $(( $1 + 100 )) python tools/fuzz_l2.py "$@" > gpurun_out/fuzz_step.log 2>&1 &
  local pid=$!
  while kill -0 $pid 2>/dev/null; do sleep 30; tail -n 1 gpurun_out/fuzz_step.log; done
  wait $pid
  grep "^fuzz seed" gpurun_out/fuzz_step.log >> $out
}
echo "## final library of round 3 (non-temporal sweep kernels in): every input planned in stretches of 128 symbols, seed 61" >> $out
LORA_CORE_PLAN=2 LORA_CORE_STRETCH=2048 LORA_FUZZ_PACKETS=24 run 300 61
echo "## defaults (short inputs on demand), seed 62" >> $out
run 200 62
echo "## default plan mode, dense streams in chunks of 64-512 symbols, seed 63" >> $out
LORA_FUZZ_PACKETS=40 LORA_FUZZ_CHUNK_SYMS=512 run 200 63
echo "## any-factor kernels, seed 64" >> $out
run 150 64 any
cat $out
