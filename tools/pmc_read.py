"""Summarise rocprofv3 --pmc CSVs: per counter, the value of the last dispatch of the demod kernel."""
import csv, glob, sys, collections
d = sys.argv[1]
out = collections.OrderedDict()
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "demod_kernel" in r["Kernel_Name"]]
    if not rows: continue
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            out[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in out.items():
    print(f"{k:34s} {v:20.0f}")
