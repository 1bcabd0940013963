"""What a GNU Radio scheduler would see: the block fed from host memory in scheduler-sized chunks (developer tool).
  python tools/stream_bench.py [sf] [fft_factor]
Dense traffic (64-symbol packets back to back) at SF `sf`; per chunk size: seconds per work() call, samples/s, launches."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gr_lora_amd as g
from importlib import import_module
synth = import_module("gr-lora_amd.synth")

if os.environ.get("LORA_SCHED_FLAG"):          # experiment: hipSetDeviceFlags (1 spin, 2 yield, 4 blocking) before the first block
    import ctypes
    import torch  # noqa: F401  (loads the HIP runtime the library uses)
    print("hipSetDeviceFlags ->", ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(int(os.environ["LORA_SCHED_FLAG"])))
import gc
gc.collect(); gc.freeze()       # a full collection of the interpreter's (torch-sized) heap is a 30 ms pause in whichever call it hits
sf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
f = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = 1 << sf
cap = synth.capture(sf, 1 << 21, symbols_per_packet=64, seed=5)
ref = None
for chunk in (1 << 21, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 4 * n):
    blk = g.demod(sf, False, 25.0, f)
    blk.work(cap[:chunk]); blk.reset()                       # warm-up: buffers
    st0 = blk.stats
    t0 = time.perf_counter()
    pdus = []
    worst = (0.0, 0)
    for k in range(0, len(cap), chunk):
        t1 = time.perf_counter()
        pdus += blk.work(cap[k:k + chunk])
        worst = max(worst, (time.perf_counter() - t1, k // chunk))
    dt = time.perf_counter() - t0
    st = blk.stats
    calls = -(-len(cap) // chunk)
    lists = [p.tolist() for p in pdus]
    if ref is None:
        ref = lists
    print(f"SF{sf} F{f} chunks of {chunk:8d} samples: {calls:6d} work() calls, {dt / calls * 1e6:9.1f} us each, {len(cap) / dt / 1e6:8.2f} Msamp/s "
          f"({len(cap) / dt / 125e3:7.0f} x a 125 kS/s channel), {st['launches'] - st0['launches']} launches, {len(pdus)} PDUs, same as one call: {lists == ref}; slowest call {worst[0] * 1e6:.0f} us (call {worst[1]})", flush=True)
    blk.close()
