#!/bin/bash
# Round-2 evidence run on the GPU box: bench at the driver's arguments, rocprofv3 kernel stats of the same command,
# HBM traffic passes, SQ counters of the headline kernel.  Results under gpurun_out/r02/.
R=$PWD; O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driverargs.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
cd $R
bash tools/hbm_traffic.sh || exit 1
bash tools/pmc.sh r02 12 2 || exit 1
python tools/pmc_read.py gpurun_out/pmc_r02 > $O/pmc_summary.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
cat $O/bench_driverargs.json | cut -c1-600; head -5 $O/kernel_stats.csv; cat $O/pmc_summary.txt
