"""Wideband capture -> channeliser -> demod block, end to end, device-resident against host buffers.
Reported beside (never as) the headline value.  SF8 at 250 kS/s inside a 1 MS/s capture, as examples/rx_usrp.grc.

  python tools/pipeline_bench.py [wideband_samples]          default 2^25 (256 MiB)"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gr_lora_amd as g
from importlib import import_module
synth = import_module("gr-lora_amd.synth")

SF, F, UP, OFF = 8, 2, 4, -0.25


def main():
    n_wb = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
    n_bb = n_wb // UP
    bb = synth.capture(SF, n_bb, symbols_per_packet=64, seed=5)
    # 4x oversample on the GPU with an ideal (FFT) interpolator, shift to the channel's offset
    x = torch.from_numpy(bb).cuda().to(torch.complex128)
    spec = torch.fft.fft(x)
    wide = torch.zeros(n_bb * UP, dtype=torch.complex128, device="cuda")
    wide[:n_bb // 2] = spec[:n_bb // 2]; wide[-(n_bb - n_bb // 2):] = spec[n_bb // 2:]
    k = torch.arange(n_bb * UP, device="cuda", dtype=torch.float64)
    ph = 2 * math.pi * torch.remainder(-OFF * k, 1.0)
    wb = (torch.fft.ifft(wide) * UP * torch.complex(torch.cos(ph), torch.sin(ph))).to(torch.complex64).contiguous()
    del x, spec, wide, k, ph
    torch.cuda.synchronize()
    wb_host = wb.cpu().numpy()

    ch = g.channelizer(2 * math.pi * OFF, 1.0 / UP)
    blk = g.demod(SF, False, 25.0, F)
    out = torch.empty(ch.output_count(len(wb)) + 8, dtype=torch.complex64, device="cuda")

    def device_path():
        ch.reset(); blk.reset()
        t0 = time.perf_counter()
        y = ch.work(wb, out=out)
        t1 = time.perf_counter()
        pdus = blk.work(y)
        t2 = time.perf_counter()
        return pdus, t1 - t0, t2 - t1

    def host_path():
        ch.reset(); blk.reset()
        t0 = time.perf_counter()
        y = ch.work(wb_host)
        t1 = time.perf_counter()
        pdus = blk.work(y)
        t2 = time.perf_counter()
        return pdus, t1 - t0, t2 - t1

    for name, fn in (("device-resident", device_path), ("host buffers", host_path)):
        fn()
        best = None
        for _ in range(3):
            pdus, tc, td = fn()
            if best is None or tc + td < best[1] + best[2]:
                best = (pdus, tc, td)
        pdus, tc, td = best
        st = blk.stats
        print(f"{name}: {len(wb) / 1e6:.1f} M wideband samples ({len(wb) * 8 / 2**20:.0f} MiB) -> {len(pdus)} PDUs; "
              f"channeliser {tc * 1e3:.2f} ms + block {td * 1e3:.2f} ms ({st['calls']} calls, {st['launches']} launches) = "
              f"{(tc + td) * 1e3:.2f} ms, {len(wb) / (tc + td) / 1e6:.0f} M wideband samples/s", flush=True)
        if name == "device-resident":
            ref = pdus
        else:
            same = len(ref) == len(pdus) and all(np.array_equal(a, b) for a, b in zip(ref, pdus))
            print("PDUs identical between the two paths:", same, flush=True)


if __name__ == "__main__":
    main()
