# throughput against resident workgroups per CU, for the libraries named on the command line
for lib in "$@"; do for b in 1 2 3; do echo "== $lib blocks/CU=$b"; LORA_HIP_LIB=$PWD/build_ab/$lib.so LORA_HIP_BLOCKS_PER_CU=$b timeout -k 10 120 python -u tools/quick_bench.py 12,2 12,1 2>&1 | grep SF; done; done
