"""GPU parity for fft_factor values other than 1, 2, 4 (the reference hands any M = N*fft_factor to FFTW,
lib/demod_impl.cc:80-81; its d_fft_size is an unsigned short, so M stays below 65536).  These run the
any-factor kernels: N-point transforms behind a run-time input twist.  The oracle takes the same sizes
through its mixed-radix transform; the bar is the one of the other parity tests."""
import os
import sys

import numpy as np
import pytest

import cases
import oracle
from test_gpu_l1_parity import check_against
from test_gpu_l2_parity import compare_with_oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu

FACTORS = [(6, 3), (6, 67), (7, 5), (7, 16), (8, 3), (8, 6), (9, 3), (9, 8), (10, 8), (10, 5), (11, 3), (11, 6),
           (12, 3), (12, 8), (12, 15)]


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


@pytest.mark.parametrize("sf,f", FACTORS)
def test_windows_against_the_oracle(sf, f, g):
    n = 1 << sf
    cap, _, _ = cases.make_capture(sf, [10], seed=500 + 13 * sf + f, snr_db=5)
    s = cases.gr_framed(cap, sf)
    rng = np.random.default_rng(sf * 100 + f)
    nwin = 96
    pos = rng.integers(0, len(s) - n, nwin).astype(np.uint64)
    off = (rng.integers(0, 16, nwin) * (n // 16)).astype(np.uint16)
    kind = rng.integers(0, 2, nwin).astype(np.uint8)
    d = oracle.Demod(sf, False, cases.BETA, f)
    ri, rv, rm = d.window_argmax(s, pos, off, kind)
    h = g.HipDemod(sf, f, cases.BETA)
    assert h.kernel_name.endswith(",any>")
    idx, val = h.batch(s, g.make_desc(pos, off, kind))
    flips, sub = check_against(idx, val, ri, rv, rm, f"SF{sf} F{f}")
    if f <= 16:
        assert sub < nwin // 2      # the comparison must not be vacuous
    else:
        # heavy oversampling puts the two bins beside a peak within 1e-4 of each other (margin ~ 1/F^2):
        # there the index is checked to sit on the same peak instead, within a fraction of one N-point bin
        m = n * f
        dist = np.abs(((idx.astype(np.int64) - ri.astype(np.int64)) + m // 2) % m - m // 2)
        assert dist.max() <= f // 8, f"SF{sf} F{f}: argmax {dist.max()} bins from the oracle's"
    # strided form == descriptor form, bit for bit
    count = (len(s) - n) // n + 1
    i1, v1 = h.batch_strided(s, 0, n, count, chirp_off=n // 8, kind=g.capi.WIN_UP)
    i2, v2 = h.batch(s, g.make_desc(np.arange(count, dtype=np.uint64) * n, n // 8, g.capi.WIN_UP))
    assert np.array_equal(i1, i2) and np.array_equal(v1, v2)
    h.close()


@pytest.mark.parametrize("sf,f", [(7, 8), (8, 3), (12, 3)])
def test_pure_tone_lands_in_bin_f_times_b(sf, f, g):
    n = 1 << sf
    _, down = oracle.chirp_tables(sf)
    bins = np.array([0, 1, n // 3, n - 1])
    k = np.arange(n)
    iq = np.concatenate([np.conj(down[:n].astype(np.complex128)) * np.exp(2j * np.pi * b * k / n) for b in bins]).astype(np.complex64)
    h = g.HipDemod(sf, f, cases.BETA)
    idx, _ = h.batch_strided(iq, 0, n, len(bins))
    assert np.array_equal(idx, (f * bins).astype(np.uint16))
    h.close()


@pytest.mark.parametrize("sf,f,ldr", [(7, 8, 0), (8, 3, 0), (9, 6, 1), (10, 3, 0)])
def test_block_against_the_oracle(sf, f, ldr, g):
    cap, _, _ = cases.make_capture(sf, [14, 9], seed=900 + sf + f, snr_db=None)
    r = compare_with_oracle(g, sf, f, ldr, cap, name=f"SF{sf} F{f}")
    assert r["pdus"] >= 1 or not r["same_path"]
    chunks = [len(cap) // 3, len(cap) // 5]
    r2 = compare_with_oracle(g, sf, f, ldr, cap, chunks=chunks, name=f"SF{sf} F{f} chunked")
    assert r2["pdus"] == r["pdus"]


def test_factor_limit_follows_the_16_bit_fft_size(g):
    with pytest.raises(g.LoraHipError) as e:
        g.HipDemod(12, 16, cases.BETA)          # M = 65536 wraps the reference's unsigned short
    assert e.value.code == g.capi.EUNSUPPORTED
    h = g.HipDemod(12, 15, cases.BETA)
    h.close()
    assert not oracle.lib().lo_demod_create(12, 0, 25.0, 16)


def test_random_captures_follow_the_oracle_at_any_fft_factor():
    import gr_lora_amd as g
    from fuzz_l2 import random_case, ANY_FACTORS
    rng = np.random.default_rng(20241)
    same = div = pdus = 0
    for k in range(150):
        sf, f, ldr, cap, chunks = random_case(rng, ANY_FACTORS)
        r = compare_with_oracle(g, sf, f, ldr, cap, chunks=chunks, name=f"fuzz-any {k} sf{sf} f{f} ldr{ldr}")
        same += r["same_path"]; div += not r["same_path"]; pdus += r["pdus"]
    print(f"{same} identical call paths, {div} sub-margin divergences, {pdus} PDUs")
    assert same >= 135 and pdus >= 100
