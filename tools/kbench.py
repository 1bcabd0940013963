"""A/B bench of the strided hot path (developer tool): settled clocks, many launches, and a bit-for-bit check of the
strided sweep against the descriptor form (which always runs the single-window kernel).
  LORA_HIP_LIB=build_ab/x.so python tools/kbench.py 12,2 12,1 [--reps 300] [--settle 0.4]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gr_lora_amd as g


def run(sf, f, reps, settle, total_bytes=1 << 30):
    n = 1 << sf
    count = total_bytes // (8 * n) + 1          # odd window counts exercise the half-filled last group
    dev = torch.device("cuda:0")
    iq = torch.randn(count * n + 77, 2, device=dev, dtype=torch.float32)
    oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
    h = g.HipDemod(sf, f)
    s = torch.cuda.Stream()
    h.set_stream(s.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < settle:
            for _ in range(20):
                h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov, async_=True)
            s.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov, async_=True)
        e1.record(s); s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rate = count / (ms * 1e-3)
    gbs = rate * (8 * n + 8) / 1e9
    # bit-for-bit against the descriptor form on a sample of windows (start, middle, the odd tail), other offset too
    ok = True
    for pos0, off in ((0, 0), (13, n // 4)):
        ncheck = min(count, 4001)
        h.batch_strided(iq, pos0, n, count, chirp_off=off, out_idx=oi, out_val=ov, async_=True)
        s.synchronize()
        sel = np.unique(np.concatenate([np.arange(ncheck // 2), np.arange(count // 2, count // 2 + ncheck // 4),
                                        np.arange(count - ncheck // 4, count)])).astype(np.uint64)
        desc = g.make_desc(sel * n + pos0, off, g.capi.WIN_UP)
        di = torch.empty(len(sel), dtype=torch.int16, device=dev); dv = torch.empty(len(sel), dtype=torch.float32, device=dev)
        h.batch(iq, torch.from_numpy(desc.view(np.uint8).reshape(-1, 16)).to(dev), out_idx=di, out_val=dv)
        a = oi.cpu().numpy().view(np.uint16)[sel.astype(np.int64)]; b = ov.cpu().numpy().view(np.uint32)[sel.astype(np.int64)]
        ok &= bool(np.array_equal(a, di.cpu().numpy().view(np.uint16)) and np.array_equal(b, dv.cpu().numpy().view(np.uint32)))
    print(f"SF{sf} F{f} [{h.kernel_name}]: {count} windows  {ms:.4f} ms  {rate/1e6:.1f} Msym/s  {100*gbs/8000:.1f}% of 8 TB/s  "
          f"strided==descriptor: {'OK' if ok else 'MISMATCH'}", flush=True)
    h.close()
    return ok


if __name__ == "__main__":
    args = sys.argv[1:]
    reps, settle = 300, 0.4
    if "--reps" in args:
        i = args.index("--reps"); reps = int(args[i + 1]); del args[i:i + 2]
    if "--settle" in args:
        i = args.index("--settle"); settle = float(args[i + 1]); del args[i:i + 2]
    gib = 1.0
    if "--gib" in args:
        i = args.index("--gib"); gib = float(args[i + 1]); del args[i:i + 2]
    cfgs = [tuple(int(x) for x in a.split(",")) for a in args] or [(12, 2), (12, 1), (12, 4)]
    good = all([run(sf, f, reps, settle, int(gib * (1 << 30))) for sf, f in cfgs])
    sys.exit(0 if good else 1)
