"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/hbm_traffic.sh into profiles/hbm_traffic.json.
gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is."""
import csv, glob, hashlib, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
kernel = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(root, "gpurun_out", "pmc_hbm", c, "**", "*counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "demod_kernel<12, 2, true" in r["Kernel_Name"] and r["Counter_Name"] == c]   # the headline kernel only
    per = [float(r["Counter_Value"]) for r in rows]
    vals[c] = sum(per) / len(per)
    kernel = rows[0]["Kernel_Name"]
line = json.loads([l for l in open(os.path.join(root, "gpurun_out", "pmc_hbm", "FETCH_SIZE.log")) if l.startswith("{")][-1])
sha = hashlib.sha256(open(os.path.join(root, "gr-lora_amd", "csrc", "lora_hip_kernels.hip.h"), "rb").read()).hexdigest()
out = {
    "kernel_header_sha256": sha,      # bench.py passes the figure on only while the kernel source is this one
    "kernel": line["config"]["kernel"], "rocprof_kernel_name": kernel,
    "windows_per_launch": line["roofline"]["symbols_per_launch"],
    "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
    "hbm_bytes_per_launch": int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024),
    "algorithmic_bytes_per_launch": line["roofline"]["symbols_per_launch"] * line["roofline"]["bytes_per_symbol"],
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py --steps 3; mean over launches; FETCH_SIZE x2 (gfx950 half-count of wide streaming reads), KiB -> bytes",
}
json.dump(out, open(os.path.join(root, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(out)
