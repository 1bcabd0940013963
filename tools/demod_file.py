#!/usr/bin/env python3
"""Demodulate a raw float32 I/Q capture file on the GPU and print one line per PDU (symbol values).

  python tools/demod_file.py capture.32iq --sf 8 --fft-factor 2 [--ldr] [--beta 25] [--device 0]
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
    a = ap.parse_args()
    import gr_lora_amd as g
    iqfile = import_module("gr-lora_amd.iqfile")
    blk = g.demod(a.sf, a.ldr, a.beta, a.fft_factor, device=a.device)
    t0 = time.perf_counter()
    pdus = iqfile.demod_file(a.path, blk)
    dt = time.perf_counter() - t0
    for p in pdus:
        print(" ".join(str(int(s)) for s in p))
    st = blk.stats
    print(f"# {len(pdus)} PDUs, {st['calls']} calls, {st['launches']} launches, {dt*1e3:.1f} ms", file=sys.stderr)
    blk.close()


if __name__ == "__main__":
    main()
