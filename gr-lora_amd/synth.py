"""Synthetic LoRa captures for benches and demos (product-side, numpy only).

Mirrors the frame layout of the reference's mod block (lib/mod_impl.cc:80-139): 4N zeros,
8 up-chirps, two sync-word chirps, 2.25 down-chirps, one shifted up-chirp per symbol,
4N+128 zeros.  Chirp tables come from the library's own builder (lora_hip_make_chirp_tables,
which restates lib/demod_impl.cc:92-101 == lib/mod_impl.cc:60-69).
"""
import numpy as np

from . import capi

NUM_PREAMBLE_CHIRPS = 8      # lib/mod_impl.h:30


def modulate(sf, symbols, sync_word=0x12):
    n = 1 << sf
    up, down = capi.make_chirp_tables(sf)
    i = np.arange(n)
    parts = [np.zeros(4 * n, np.complex64)]
    parts += [up[i % n]] * NUM_PREAMBLE_CHIRPS
    parts.append(up[(8 * ((sync_word & 0xF0) >> 4) + i) % n])
    parts.append(up[(8 * (sync_word & 0x0F) + i) % n])
    parts.append(down[np.arange(2 * n + n // 4) % n])
    for s in np.asarray(symbols, np.int64):
        parts.append(up[(s + n // 4 + i) % n])
    parts.append(np.zeros(4 * n + 128, np.complex64))
    return np.concatenate(parts).astype(np.complex64)


def capture(sf, n_samples, symbols_per_packet=64, seed=2, sync_word=0x12):
    """Back-to-back packets (with the mod's own zero gaps) truncated/padded to n_samples."""
    rng = np.random.default_rng(seed)
    n = 1 << sf
    out, total = [], 0
    while total < n_samples:
        syms = rng.integers(0, n, symbols_per_packet)
        syms[:8] &= ~3
        f = modulate(sf, syms, sync_word)
        out.append(f); total += len(f)
    return np.concatenate(out)[:n_samples]
