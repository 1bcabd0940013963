#!/bin/bash
# The bench line of record and rocprofv3 kernel stats of the same command (run after tools/r03_profile.sh has refreshed the
# counter files, so that the line carries them): results under gpurun_out/r03b/
R=$PWD; O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driverargs.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/under_rocprof.json 2> $O/rocprof.err || exit 1
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
head -c 300 $O/bench_driverargs.json; echo; head -2 $O/kernel_stats.csv | tail -1
