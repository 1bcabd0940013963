#!/bin/bash
# rocprofv3 kernel stats and HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes) for the channeliser's decimation kernel at the rx_usrp.grc setting.
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/chan_prof; mkdir -p $R/gpurun_out/chan_prof
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chan_prof/stats -- python3 $R/tools/chan_prof.py 0.25 -0.25 20 > $R/gpurun_out/chan_prof/stats.log 2>&1 || { echo "stats failed"; tail -3 $R/gpurun_out/chan_prof/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/chan_prof/$c -- python3 $R/tools/chan_prof.py 0.25 -0.25 3 > $R/gpurun_out/chan_prof/$c.log 2>&1 || { echo "$c failed"; tail -3 $R/gpurun_out/chan_prof/$c.log; exit 1; }
done
find $R/gpurun_out/chan_prof -name "*.csv" | head -20
