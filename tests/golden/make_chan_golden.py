#!/usr/bin/env python3
"""Generate tests/golden/chan_golden.npz: known answers of the channeliser oracle (oracle/chan_oracle.c).

The reference holds no fixture for the two GNU Radio blocks in front of lora.demod (examples/rx_usrp.grc:489-503,
:825-860) and GNU Radio is not importable here, so these vectors are produced by the CPU oracle and are "parity
unpinned" with respect to GNU Radio itself.  They pin the oracle against regressions and give the HIP path fixed
expected outputs that do not depend on the oracle being compiled on the GPU box.

  python tests/golden/make_chan_golden.py

Per case c (rate, rotation offset in cycles/sample, own prototype or designed taps; tests/cases.py is not involved):
  c<k>_rate, c<k>_offset, c<k>_taps      parameters and the prototype filter
  c<k>_out                               the float oracle's output for the shared input `x` (rotator -> resampler)
  c<k>_pos, c<k>_acc                     the index recurrence: 32*i_in + arm and d_acc per output
  c<k>_n                                 how many samples of the shared input `x` the case uses
and `x` (2048 seeded complex64 samples), `taps_sha256_<rate>` of the designed prototypes.
"""
import hashlib
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

CASES = [(0.25, 0.0, None), (0.25, -0.25, None), (0.125, 0.1837, None), (0.3, 0.0, None), (0.7, 0.0, None),
         (0.3, -0.41, None), (1.5, 0.0, "own"), (2.75, 0.0, "own"), (0.4, 0.0, None), (40.0, 0.0, "own")]


def own_taps():
    return (np.hamming(32 * 12 + 5) * np.sinc(np.linspace(-6, 6, 32 * 12 + 5))).astype(np.float32)


def main():
    rng = np.random.default_rng(20261110)
    x = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    out = {"x": x}
    for k, (rate, off, taps) in enumerate(CASES):
        taps = own_taps() if taps == "own" else oracle.design_taps(rate)
        n = len(x) if rate < 8 else 128                    # keep the fixture small: rate 40 makes 40 outputs per input
        xr = oracle.Rotator(2 * math.pi * off).work(x[:n]) if off else x[:n]
        y = oracle.PfbArbResampler(rate, taps).work(xr)
        pos, acc = oracle.PfbArbResampler(rate, taps).schedule(n)
        assert len(pos) == len(y)
        out.update({f"c{k}_rate": np.float64(rate), f"c{k}_offset": np.float64(off), f"c{k}_taps": taps, f"c{k}_n": np.int64(n),
                    f"c{k}_out": y, f"c{k}_pos": pos.astype(np.uint32), f"c{k}_acc": acc})
    for rate in (0.5, 0.25, 0.125, 0.0625, 0.3):
        out[f"taps_sha256_{rate}"] = np.frombuffer(hashlib.sha256(oracle.design_taps(rate).tobytes()).digest(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "chan_golden.npz"), **out)
    print("wrote chan_golden.npz:", len(CASES), "cases,", os.path.getsize(os.path.join(HERE, "chan_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
