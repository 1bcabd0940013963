#!/bin/bash
# Collect FETCH_SIZE and WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) for bench.py's kernel.
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pmc_hbm; mkdir -p $R/gpurun_out/pmc_hbm
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_hbm/$c -- python3 $R/bench.py --steps 3 --warmup 1 --settle-s 0 --no-cpu-baseline > $R/gpurun_out/pmc_hbm/$c.log 2>&1 || { echo "$c failed"; tail -3 $R/gpurun_out/pmc_hbm/$c.log; exit 1; }
done
