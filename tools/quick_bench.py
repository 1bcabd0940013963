"""Quick throughput sweep of the strided hot path with IQ resident in HBM (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gr_lora_amd as g

def run(sf, f, total_bytes=1 << 30, reps=5):
    n = 1 << sf
    count = total_bytes // (8 * n)
    dev = torch.device("cuda:0")
    iq = torch.randn(count * n, 2, device=dev, dtype=torch.float32)
    oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
    h = g.HipDemod(sf, f)
    s = torch.cuda.Stream()
    h.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
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
    print(f"SF{sf} F{f}: {count} windows  {ms:.3f} ms  {rate/1e6:.1f} Msym/s  {gbs:.0f} GB/s  {100*gbs/8000:.1f}% of 8 TB/s", flush=True)
    h.close()

if __name__ == "__main__":
    cfgs = [(12, 2), (12, 1), (12, 4), (11, 2), (10, 2), (9, 2), (8, 2), (7, 2)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for sf, f in cfgs:
        run(sf, f)
