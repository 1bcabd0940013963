"""Write profiles/valu_count.json from a tools/pmc_read.py summary of the headline kernel (tools/pmc.sh r03 12 2) and,
optionally, the LORA_ABL=33 timing (tools/kbench.py line of the build without loads and LDS exchanges).
  python tools/valu_count.py gpurun_out/r03/pmc_summary.txt [gpurun_out/r03/abl33.log]
The file names the SHA-256 of the kernel source it was measured on; bench.py withholds the figures when that changes."""
import hashlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) == 2:
        vals[p[0]] = float(p[1])
windows = 32768                      # tools/prof_one.py 12 2: 1 GiB / 32 KiB
sha = hashlib.sha256(open(os.path.join(root, "gr-lora_amd", "csrc", "lora_hip_kernels.hip.h"), "rb").read()).hexdigest()
out = {
    "kernel_header_sha256": sha,
    "kernel": "demod_kernel<12,2>",
    "valu_wave_instr_per_symbol": vals["SQ_INSTS_VALU"] / windows,
    "source": f"rocprofv3 --pmc SQ_INSTS_VALU on tools/prof_one.py 12 2 ({windows} windows per launch): {int(vals['SQ_INSTS_VALU'])} per launch ({os.path.relpath(sys.argv[1], root)})",
    "peak_spec_g_wave_instr_per_s": 1228.8,
    "peak_note": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 f32 instruction (= 157.3 TFLOP/s fp32 FMA); under this kernel's load the part holds 1.9 GHz (in-kernel clock, profiles/r02_kernel_experiments.md), i.e. ~970 G/s deliverable",
    "lds_bytes_per_symbol": 294912,
    "lds_peak_gbs": 78643.2,
    "lds_note": "LDS roof (SURVEY 8d): bytes through LDS per symbol = 2 sub-transforms x 256 threads x (16+16 exchange 1, 8 pass-1 twiddles, 16+16 exchange 2) x 8 B; peak = 256 CUs x 128 B/clk x 2.4 GHz; measured SQ_LDS_IDX_ACTIVE "
                f"{vals.get('SQ_LDS_IDX_ACTIVE', 0) / 1e6:.1f} M cycles per launch",
}
if len(sys.argv) > 2:
    m = re.search(r"SF12 F2 .*?(\d+) windows\s+([\d.]+) ms", open(sys.argv[2]).read())
    out["valu_only_ms_per_launch"] = float(m.group(2))
    out["valu_only_windows_per_launch"] = int(m.group(1))
    out["valu_only_note"] = ("the same kernel built with LORA_ABL=33 (no IQ loads, no LDS exchanges and their barriers: wrong results, timing only), "
                             "tools/kbench.py in the same gpurun call as the counters: what a perfectly hidden memory system would leave")
json.dump(out, open(os.path.join(root, "profiles", "valu_count.json"), "w"), indent=1)
print(out)
