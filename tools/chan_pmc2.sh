#!/bin/bash
# SQ counters of the channeliser kernels (what binds them): tools/chan_pmc2.sh <rate> <offset> <tag>
rate=$1; off=$2; tag=$3
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf $R/gpurun_out/chan_pmc_$tag; mkdir -p $R/gpurun_out/chan_pmc_$tag
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/chan_pmc_$tag/p$i -- python3 $R/tools/chan_prof.py $rate $off 2 > $R/gpurun_out/chan_pmc_$tag/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/chan_pmc_$tag/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
out = collections.OrderedDict()
name = None
for f in sorted(glob.glob("$R/gpurun_out/chan_pmc_$tag/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "lora_chan::chan_" in r["Kernel_Name"] and "carry" not in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            out[r["Counter_Name"]] = float(r["Counter_Value"]); name = r["Kernel_Name"]
print("kernel:", name, " rate $rate offset $off, 1 GiB per launch")
for k, v in out.items():
    print(f"{k:28s} {v:18.0f}")
PY
