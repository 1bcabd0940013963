#!/bin/bash
# Host code under AddressSanitizer (tools/asan_driver.cpp).  Build here (no GPU needed), run on the GPU box:
#   bash tools/asan_run.sh build                      -> build_ab/asan.so, build_ab/asan_driver
#   gpurun --timeout 600 -- 'bash tools/asan_run.sh run'
set -e
cd "$(dirname "$0")/.."
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O1 -g -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -Wall -Wno-unused-result -Iinclude"
if [ "$1" = build ]; then
  mkdir -p build_ab
  ( cd gr-lora_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 ${FLAGS//-Iinclude/-I../../include} -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
      -shared -o ../../build_ab/asan.so lora_hip_demod.hip lora_hip_chan.hip demod_core.cpp decode_chain.cpp )
  $CLANG -O1 -g -std=c++17 -fsanitize=address -fno-omit-frame-pointer -Iinclude tools/asan_driver.cpp build_ab/asan.so \
      -Wl,-rpath,'$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o build_ab/asan_driver
  echo built
  exit 0
fi
out=gpurun_out/r03_host_asan.txt
mkdir -p gpurun_out; : > $out
export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:abort_on_error=0:halt_on_error=1
for cfg in "8 2 700" "9 2 300" "7 4 900" "10 1 150"; do
  set -- $cfg
  python -c "
import sys; sys.path.insert(0, '.')
import gr_lora_amd as g
from gr_lora_amd import synth
synth.capture($1, $3 * (1 << $1), symbols_per_packet=24, seed=9).tofile('gpurun_out/asan_cap.c64')"
  echo "## SF$1 fft_factor $2, $3 symbol times of back-to-back 24-symbol packets" | tee -a $out
  timeout -k 10 120 build_ab/asan_driver gpurun_out/asan_cap.c64 $1 $2 2>&1 | tee -a $out
  echo "## the same in stretches of 128 symbols (LORA_CORE_STRETCH=2048)" | tee -a $out
  LORA_CORE_STRETCH=2048 timeout -k 10 120 build_ab/asan_driver gpurun_out/asan_cap.c64 $1 $2 2>&1 | tee -a $out
done
rm -f gpurun_out/asan_cap.c64
