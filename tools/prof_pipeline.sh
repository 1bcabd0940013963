# rocprofv3 kernel stats of tools/pipeline_bench.py (channeliser -> planned demod block)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pipe && mkdir -p $R/gpurun_out/prof_pipe
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pipe -- python3 $R/tools/pipeline_bench.py > $R/gpurun_out/prof_pipe.log 2>&1 || { echo failed; tail -5 $R/gpurun_out/prof_pipe.log; exit 1; }
cd $R && python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_pipe/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["TotalDurationNs"])
PY
grep "device-resident\|host buffers" gpurun_out/prof_pipe.log
