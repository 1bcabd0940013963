"""The oracle at FFT sizes that are not powers of two (fft_factor 3, 5, 6 ...): its mixed-radix transform
against the independent numpy recomputation, and the block level through a loopback."""
import numpy as np
import pytest

import cases
import oracle
from oracle import numpy_check as nc


@pytest.mark.parametrize("sf,f", [(6, 3), (6, 67), (7, 5), (7, 7), (8, 6), (8, 3), (9, 12), (10, 3), (12, 3), (12, 15), (8, 8)])
def test_window_spectrum_peak_agrees_with_numpy(sf, f):
    n = 1 << sf
    rng = np.random.default_rng(sf * 31 + f)
    iq = (rng.standard_normal(6 * n) + 1j * rng.standard_normal(6 * n)).astype(np.complex64)
    d = oracle.Demod(sf, False, cases.BETA, f)
    for kind in (oracle.KIND_UP, 1):
        pos = np.array([0, n // 2, 3 * n + 5], np.uint64)
        idx, val, margin = d.window_argmax(iq, pos, n // 4, kind)
        for i, p in enumerate(pos):
            m = nc.window_spectrum_np(iq, int(p), n // 4, kind, sf, f)
            assert abs(float(val[i]) - m.max()) <= 2e-6 * m.max()
            if margin[i] > 1e-4:
                assert int(idx[i]) == int(np.argmax(m))


@pytest.mark.parametrize("sf,f", [(7, 3), (8, 6), (9, 8)])
def test_loopback_demodulates_at_odd_factors(sf, f):
    cap, tx, _ = cases.make_capture(sf, [12], seed=70 + sf + f, snr_db=None)
    d = oracle.Demod(sf, False, cases.BETA, f)
    pdus, _, n_calls, n_ffts = d.stream(cap)
    assert len(pdus) == 1 and n_ffts > n_calls
    got = np.asarray(pdus[0][:len(tx[0])]).astype(np.int64)
    want = np.asarray(tx[0]).astype(np.int64)
    want[:8] //= 4                                   # header symbols come out divided by 4 (:363-387)
    n = 1 << sf
    off = np.minimum((got - want) % n, (want - got) % n)
    assert np.all(off[8:] <= 2) and np.all(off[:8] <= 1)   # the reference's own +-1/2 symbol slips


def test_size_limit():
    assert not oracle.lib().lo_demod_create(12, 0, 25.0, 16)      # 65536 wraps d_fft_size
    h = oracle.Demod(12, False, cases.BETA, 15)
    del h
