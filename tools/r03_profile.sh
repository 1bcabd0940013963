#!/bin/bash
# Round-3 evidence run on the GPU box: bench at the driver's arguments, rocprofv3 kernel stats of the same command,
# HBM traffic passes, SQ counters of the headline kernel, the VALU-only build (build_ab/abl33.so, built beforehand with
# -DLORA_ABL=33) timed beside the product in the same call, kernel stats of the channeliser.  Results under gpurun_out/r03/.
R=$PWD; O=$R/gpurun_out/r03; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driverargs.json 2> $O/bench.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err || exit 1
cd $R
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
bash tools/hbm_traffic.sh || exit 1
bash tools/pmc.sh r03 12 2 || exit 1
python tools/pmc_read.py gpurun_out/pmc_r03 > $O/pmc_summary.txt || exit 1
if [ -f build_ab/abl33.so ]; then
  timeout -k 10 150 python -u tools/kbench.py 12,2 > $O/kbench_product.log 2>&1 || exit 1
  # (its results are wrong by construction, so kbench's strided == descriptor check fails and it exits non-zero)
  LORA_HIP_LIB=$R/build_ab/abl33.so timeout -k 10 150 python -u tools/kbench.py 12,2 > $O/abl33.log 2>&1; grep -q "SF12 F2" $O/abl33.log || exit 1
fi
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chan_stats -- python3 $R/tools/chan_prof.py 0.3 -0.25 20 > $O/chan_prof.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chan_stats_decim -- python3 $R/tools/chan_prof.py 0.25 -0.25 20 > $O/chan_prof_decim.log 2>&1 || exit 1
cd $R
find $O/chan_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/chan_kernel_stats.csv
find $O/chan_stats_decim -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/chan_decim_kernel_stats.csv
rm -rf $O/stats $O/chan_stats $O/chan_stats_decim
cut -c1-700 $O/bench_driverargs.json; head -4 $O/kernel_stats.csv; cat $O/pmc_summary.txt; grep SF $O/kbench_product.log $O/abl33.log; head -4 $O/chan_kernel_stats.csv
