"""Experiment: one batch of SF12 F=2 windows swept by k concurrent launches on k streams (k handles) against one launch.
  python tools/two_stream_bench.py [k ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_lora_amd as g

sf, f = 12, 2
n = 1 << sf
count = (1 << 30) // (8 * n)
dev = torch.device("cuda:0")
iq = torch.randn(count * n + 77, 2, device=dev, dtype=torch.float32)
oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
for k in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 1, 2]:
    hs = [g.HipDemod(sf, f) for _ in range(k)]
    ss = [torch.cuda.Stream() for _ in range(k)]
    for h, s in zip(hs, ss):
        h.set_stream(s.cuda_stream)
    per = count // k
    def step():
        for j, h in enumerate(hs):
            lo = j * per
            c = per if j < k - 1 else count - lo
            h.batch_strided(iq, lo * n, n, c, out_idx=oi[lo:lo + c], out_val=ov[lo:lo + c], async_=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{k} concurrent launches per batch: {dt * 1e3:.4f} ms per batch, {count / dt / 1e6:.1f} Msym/s", flush=True)
    for h in hs:
        h.close()
