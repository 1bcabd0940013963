"""Throughput of the channeliser (rotator + polyphase resampler) with the wideband capture resident in HBM.

  python tools/chan_bench.py [rate,offset[,general] ...]       default: the rx_usrp.grc case (0.25, -0.25) and others

One step = one lora_hip_chan_work call over 1 GiB of complex64 (134 M samples).  Algorithmic bytes per input
sample = 8 (read once) + 8*rate (output written once); flops per input sample = 2*2*taps_per_filter*rate
(complex sample x real tap, one arm per output).  Developer tool; the CPU leg times the oracle's restatement of
the two GNU Radio blocks on one core over a bounded sample of the same stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gr_lora_amd as g


def run(rate, offset, general=False, total_bytes=1 << 30, reps=10, cpu=True):
    dev = torch.device("cuda:0")
    n = total_bytes // 8
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, dtype=torch.float32))
    ch = g.channelizer(2 * np.pi * offset, rate)
    if general:
        ch.force_general(True)
    out = torch.empty(ch.output_count(n) + 8, dtype=torch.complex64, device=dev)
    s = torch.cuda.Stream()
    ch.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        for _ in range(3):
            ch.work(x, out=out, async_=True)
        s.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            y = ch.work(x, out=out, async_=True)
        e1.record(s); s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    k = ch.taps_per_filter
    gbs = (8 * n + 8 * len(y)) / (ms * 1e-3) / 1e9
    flop = 4.0 * k * len(y) * (1 if ch.flt_rate == 0 else 2)
    line = (f"rate {rate:g} offset {offset:g} {ch.kernel_name} (K={k}): {n / 1e6:.0f} M in -> {len(y) / 1e6:.1f} M out  "
            f"{ms:.3f} ms  {n / ms / 1e6:.2f} Gsamp/s  {gbs:.0f} GB/s = {100 * gbs / 8000:.1f}% of 8 TB/s  "
            f"{flop / ms / 1e9:.1f} TFLOP/s")
    if cpu:
        import oracle     # CPU baseline leg only
        m = 2_000_000
        xs = x[:m].cpu().numpy()
        rot, pfb = oracle.Rotator(2 * np.pi * offset), oracle.PfbArbResampler(rate)
        t0 = time.perf_counter()
        pfb.work(rot.work(xs))
        dt = time.perf_counter() - t0
        line += f"  | CPU oracle, 1 core, {m / 1e6:.0f} M samples: {m / dt / 1e6:.1f} Msamp/s"
    print(line, flush=True)
    ch.close()


if __name__ == "__main__":
    cfgs = [(0.25, -0.25, False), (0.125, 0.1837, False), (0.5, 0.05, False), (0.0625, 0.3, False), (0.25, 0.0, False),
            (0.25, -0.25, True), (0.3, -0.41, False)]
    if len(sys.argv) > 1:
        cfgs = []
        for a in sys.argv[1:]:
            p = a.split(",")
            cfgs.append((float(p[0]), float(p[1]), len(p) > 2))
    for rate, off, gen in cfgs:
        run(rate, off, gen, reps=3 if (gen or rate not in (0.5, 0.25, 0.125, 0.0625)) else 10)
