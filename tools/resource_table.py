"""Per-variant register/scratch table of the demod kernels (developer tool, CPU only: hipcc cross-compiles).
  python tools/resource_table.py [extra hipcc -D flags ...] > profiles/r03_resource_usage.md
Runs `make resource-usage` (-Rpass-analysis=kernel-resource-usage) and prints one row per demod_kernel instantiation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(extra=()):
    cmd = ["make", "-s", "-C", os.path.join(ROOT, "gr-lora_amd", "csrc"), "resource-usage"]
    if extra:
        cmd.append("EXTRA=" + " ".join(extra))
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def demangle(names):
    """demod_kernelILi12ELi2ELb1ELb0EE -> demod_kernel<12, 2, true, false>; other kernels: the name between the length prefixes"""
    out = []
    for n in names:
        m = re.search(r"demod_kernelILi(\d+)ELi(\d+)ELb([01])ELb([01])E(?:Lb([01])E)?E", n)
        if m:
            b = ("false", "true")
            out.append(f"demod_kernel<{m.group(1)}, {m.group(2)}, {b[int(m.group(3))]}, {b[int(m.group(4))]}>" + (" nt" if m.group(5) == "1" else ""))
        else:
            k = re.findall(r"\d+([a-z_][a-z_0-9]*kernel[a-z_0-9]*)", n)
            out.append(k[0] if k else n)
    return out


def main():
    rows = collect(sys.argv[1:])
    names = demangle([r["name"] for r in rows])
    print("| kernel | SF | F | taps in VGPRs | any-F | VGPRs | AGPRs | spilled VGPRs | scratch B/lane | SGPRs | waves/SIMD | LDS B/WG (static) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        m = re.search(r"demod_kernel<(\d+), (\d+), (true|false), (true|false)>", n)
        if m:
            sf, f, co, af = m.groups()
            short = "demod_kernel" + (" (nt loads)" if n.endswith(" nt") else "")
        else:
            sf = f = co = af = ""
            short = re.sub(r"\(.*", "", n).replace("void ", "").replace("(anonymous namespace)::", "")
        print(f"| `{short}` | {sf} | {f} | {co} | {af} | {r.get('VGPRs','')} | {r.get('AGPRs','')} | {r.get('VGPRs Spill','')} | "
              f"{r.get('ScratchSize [bytes/lane]','')} | {r.get('TotalSGPRs','')} | {r.get('Occupancy [waves/SIMD]','')} | "
              f"{r.get('LDS Size [bytes/block]','')} |")


if __name__ == "__main__":
    main()
