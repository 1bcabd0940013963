#!/usr/bin/env python3
"""Worst-case agreement of (argmax, max |X|^2) between the GPU path and the oracle, per (SF, fft_factor):
random-noise windows, clean chirp windows and noisy chirp windows, up path and SFD path, random chirp offsets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gr_lora_amd as g
import oracle, cases

def main():
    rng = np.random.default_rng(7)
    print("SF F : windows | idx mismatches (all below the 1e-4 margin?) | max rel. error of max|X|^2 | smallest margin with equal idx")
    for sf in range(6, 13):
        n = 1 << sf
        n_win = 512 if sf <= 10 else 160
        for f in (1, 2, 4):
            d = oracle.Demod(sf, False, cases.BETA, f)
            syms = oracle.random_symbols(sf, n_win // 2 + 8, 99 + sf)
            chirps = oracle.modulate(sf, syms, cases.SYNC_WORD)
            noise = (rng.standard_normal(n_win * n) + 1j * rng.standard_normal(n_win * n)).astype(np.complex64)
            iq = noise.copy()
            k = min(len(chirps), len(iq) // 2)
            iq[:k] = chirps[:k] + 0.3 * noise[:k]                      # first half: chirps in noise
            iq[k:k + k // 2] = chirps[:k // 2]                          # then clean chirps
            pos = np.sort(rng.integers(0, len(iq) - n, n_win)).astype(np.uint64)
            off = (rng.integers(0, 16, n_win) * (n // 16)).astype(np.uint16)
            kind = rng.integers(0, 2, n_win).astype(np.uint8)
            ridx, rval, margin = d.window_argmax(iq, pos, off, kind)
            h = g.HipDemod(sf, f, cases.BETA)
            idx, val = h.batch(iq, g.make_desc(pos, off, kind))
            h.close()
            mism = np.flatnonzero(idx != ridx)
            ok = bool(np.all(margin[mism] <= 1e-4))
            nz = rval > 0
            rel = float(np.max(np.abs(val[nz] - rval[nz]) / rval[nz]))
            same = idx == ridx
            print(f"SF{sf} F{f}: {n_win} | {len(mism)} ({'yes' if ok else 'NO'}) | {rel:.2e} | {margin[same].min():.2e}", flush=True)
            assert ok and rel <= 1e-4

if __name__ == "__main__":
    main()
