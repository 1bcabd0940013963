"""CPU tests of the channeliser oracle (oracle/chan_oracle.c: GNU Radio's rotator_cc and pfb_arb_resampler_ccf
restated, examples/rx_usrp.grc:489-503, :825-860) against an independent numpy double-precision
recomputation, and of the host-side pieces of the product that need no GPU (tap design, exported symbols).
PARITY UNPINNED: the reference holds no fixture for these blocks."""
import math

import numpy as np
import pytest

import oracle


def noise(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def numpy_pfb(x, rate, taps, nfilts=32):
    """pfb_arb_resampler_ccf in numpy: float recurrence for (arm, d_acc), double dot products."""
    ntaps = len(taps)
    K = -(-ntaps // nfilts)
    t = np.zeros(nfilts * K); t[:ntaps] = taps
    d = np.zeros(nfilts * K); d[:ntaps - 1] = np.diff(np.asarray(taps, np.float32)).astype(np.float32)
    arms = t.reshape(K, nfilts).T[:, ::-1]          # arms[j][k] = taps[j + (K-1-k)*nfilts]
    darms = d.reshape(K, nfilts).T[:, ::-1]
    f32 = np.float32
    dec = int(math.floor(f32(nfilts) / f32(rate)))
    flt = f32(f32(nfilts) / f32(rate)) - f32(dec)
    xin = np.concatenate([np.zeros(K - 1), x.astype(np.complex128)])
    out = []
    j, acc, i_in = (ntaps // 2) % nfilts, f32(0), 0
    while i_in < len(x):
        while j < nfilts:
            seg = xin[i_in:i_in + K]
            out.append(np.dot(seg, arms[j]) + float(acc) * np.dot(seg, darms[j]))
            acc = f32(acc + flt)
            j += dec + int(math.floor(acc))
            acc = f32(math.fmod(acc, 1.0))
        i_in += j // nfilts
        j %= nfilts
    return np.array(out)


@pytest.mark.parametrize("rate", [0.25, 0.125, 0.3, 0.7, 1.0 / 3, 0.0371])
def test_resampler_against_numpy(rate):
    x = noise(6000, 5)
    r = oracle.PfbArbResampler(rate)
    want = numpy_pfb(x, rate, r.taps)
    got_d = oracle.PfbArbResampler(rate).work(x, use_double=True)
    got_f = r.work(x)
    assert len(got_d) == len(want) == len(got_f)
    rms = np.sqrt(np.mean(np.abs(want) ** 2))
    assert np.abs(got_d - want).max() / rms <= 2e-7          # double sums rounded to float once
    assert np.abs(got_f - want).max() / rms <= 5e-6          # sequential float sums


def test_resampler_with_own_taps_interpolating():
    taps = (np.hamming(32 * 9 + 3) * np.sinc(np.linspace(-4, 4, 32 * 9 + 3))).astype(np.float32)
    x = noise(1500, 9)
    for rate in (1.0, 1.5, 2.75):
        got = oracle.PfbArbResampler(rate, taps).work(x, use_double=True)
        want = numpy_pfb(x, rate, taps)
        assert len(got) == len(want)
        assert np.abs(got - want).max() <= 3e-7 * np.sqrt(np.mean(np.abs(want) ** 2))


def test_chunking_does_not_change_the_stream():
    x = noise(20_000, 2)
    for rate in (0.25, 0.3, 1.5):
        taps = None if rate < 1 else (np.hamming(300) * np.sinc(np.linspace(-4, 4, 300))).astype(np.float32)
        whole = oracle.PfbArbResampler(rate, taps).work(x)
        r = oracle.PfbArbResampler(rate, taps)
        parts = [r.work(x[:1]), r.work(x[1:3]), r.work(x[3:7000]), r.work(x[7000:7001]), r.work(x[7001:])]
        assert np.array_equal(np.concatenate(parts), whole)
        # the schedule alone reproduces the positions work() used
        pos, acc = oracle.PfbArbResampler(rate, taps).schedule(len(x))
        assert len(pos) == len(whole) and np.all(np.diff(pos.astype(np.int64)) > 0) and np.all((acc >= 0) & (acc < 1))


def test_designed_taps():
    """firdes.low_pass_2 with a Blackman-harris window: odd length, symmetric, DC gain = nfilts, and the length
    formula of compute_ntaps_windes."""
    for rate, n in ((0.25, 2909), (0.125, 5819), (0.5, 1455)):
        t = oracle.design_taps(rate)
        assert len(t) == n and len(t) % 2 == 1
        assert np.array_equal(t, t[::-1])
        assert abs(float(t.astype(np.float64).sum()) - 32.0) < 1e-3
        # stop band: beyond cutoff + transition the response is >= 90 dB below the pass band
        h = np.abs(np.fft.rfft(t.astype(np.float64), 1 << 18)) / 32.0
        f = np.arange(len(h)) / (1 << 18) * 32.0          # in units of the input rate
        assert 20 * np.log10(h[f > 0.6 * rate + 1e-3].max()) < -90


def test_rotator_tracks_the_exact_phase():
    """The float product chain stays within 1e-8 rad/sample of exp(i*n*angle(incr)) and keeps unit magnitude."""
    n = 100_000
    for off in (-0.25, 0.1837, 0.05):
        inc = 2 * math.pi * off
        y = oracle.Rotator(inc).work(np.ones(n, np.complex64))
        assert np.abs(np.abs(y) - 1).max() < 2e-5
        ph = np.unwrap(np.angle(y * np.exp(-1j * float(np.float32(inc)) * np.arange(n))))
        assert abs(ph[-1]) / n < 1e-7
        # state carries across calls
        r = oracle.Rotator(inc)
        assert np.array_equal(np.concatenate([r.work(np.ones(777, np.complex64)), r.work(np.ones(n - 777, np.complex64))]), y)


def test_product_tap_design_matches_the_oracle_and_symbols_are_exported():
    import gr_lora_amd as g
    for rate in (0.25, 0.125, 0.3, 0.7):
        assert np.array_equal(g.create_taps(rate), oracle.design_taps(rate))
    with pytest.raises(g.LoraHipError):
        g.create_taps(1.0)
    for name in g.capi.EXPORTS:
        assert hasattr(g.capi.lib(), name), name


def _golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chan_golden.npz"), allow_pickle=False)


def test_oracle_against_committed_golden_vectors():
    """tests/golden/chan_golden.npz (made by tests/golden/make_chan_golden.py): the oracle still produces, bit for
    bit, the outputs, positions and weights it was recorded with; designed taps by hash."""
    import hashlib
    gold = _golden()
    x = gold["x"]
    n_cases = sum(1 for k in gold.files if k.endswith("_rate"))
    assert n_cases >= 10
    for k in range(n_cases):
        rate, off, taps = float(gold[f"c{k}_rate"]), float(gold[f"c{k}_offset"]), gold[f"c{k}_taps"]
        xk = x[:int(gold[f"c{k}_n"])]
        xr = oracle.Rotator(2 * math.pi * off).work(xk) if off else xk
        assert np.array_equal(oracle.PfbArbResampler(rate, taps).work(xr), gold[f"c{k}_out"]), k
        pos, acc = oracle.PfbArbResampler(rate, taps).schedule(len(xk))
        assert np.array_equal(pos, gold[f"c{k}_pos"]) and np.array_equal(acc, gold[f"c{k}_acc"]), k
    for rate in (0.5, 0.25, 0.125, 0.0625, 0.3):
        want = bytes(gold[f"taps_sha256_{rate}"])
        assert hashlib.sha256(oracle.design_taps(rate).tobytes()).digest() == want
