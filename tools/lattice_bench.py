"""Lattice sweep rate (stride N/16, chirp offset 0, register-resident taps: stage A of the planned block) for a few SF/F:
developer tool.  python tools/lattice_bench.py [sf,f ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_lora_amd as g


def sweep(sf, f, total_bytes=256 << 20, reps=5):
    n = 1 << sf
    ns = total_bytes // 8
    dev = torch.device("cuda:0")
    iq = torch.randn(ns, 2, device=dev, dtype=torch.float32)
    count = (ns - n) // (n // 16) + 1
    oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
    h = g.HipDemod(sf, f)
    s = torch.cuda.Stream(); h.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        h.batch_strided(iq, 0, n // 16, count, 0, 0, g.capi.WIN_UP, out_idx=oi, out_val=ov, async_=True); s.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            h.batch_strided(iq, 0, n // 16, count, 0, 0, g.capi.WIN_UP, out_idx=oi, out_val=ov, async_=True)
        e1.record(s); s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"lattice sweep SF{sf} F{f}: {count} overlapped windows (stride N/16) {ms:.3f} ms -> {count / ms / 1e3:.1f} Mwin/s", flush=True)
    h.close()


for sf, f in [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(8, 2), (9, 2), (10, 2), (12, 2)]:
    sweep(sf, f)
