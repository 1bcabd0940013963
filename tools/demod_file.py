#!/usr/bin/env python3
"""Demodulate a raw float32 I/Q capture file on the GPU and print one line per PDU (symbol values).

  python tools/demod_file.py capture.32iq --sf 8 --fft-factor 2 [--ldr] [--beta 25] [--device 0]
  python tools/demod_file.py wideband.32iq --sf 8 --samp-rate 1e6 --bw 250e3 --offset -250e3     # channelise first

With --samp-rate the file is a wideband capture: it is shifted by --offset and resampled to --bw on the GPU
(examples/rx_usrp.grc:95,270,401: rotator phase_inc = 2*pi*offset/samp_rate, resampler rate = bw/samp_rate).
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--sf", type=int, default=8)
    ap.add_argument("--fft-factor", type=int, default=2)
    ap.add_argument("--beta", type=float, default=25.0)
    ap.add_argument("--ldr", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--samp-rate", type=float, default=None, help="sample rate of the file; enables the channeliser")
    ap.add_argument("--bw", type=float, default=125e3, help="LoRa bandwidth = sample rate the block needs")
    ap.add_argument("--offset", type=float, default=0.0, help="channel centre relative to the capture centre, Hz")
    a = ap.parse_args()
    import gr_lora_amd as g
    iqfile = import_module("gr-lora_amd.iqfile")
    blk = g.demod(a.sf, a.ldr, a.beta, a.fft_factor, device=a.device)
    t0 = time.perf_counter()
    ch = None
    if a.samp_rate:
        import math
        ch = g.channelizer(2 * math.pi * a.offset / a.samp_rate, a.bw / a.samp_rate, device=a.device)
    pdus = iqfile.demod_file(a.path, blk, channelizer=ch)
    dt = time.perf_counter() - t0
    for p in pdus:
        print(" ".join(str(int(s)) for s in p))
    st = blk.stats
    print(f"# {len(pdus)} PDUs, {st['calls']} calls, {st['launches']} launches, {dt*1e3:.1f} ms", file=sys.stderr)
    blk.close()


if __name__ == "__main__":
    main()
