"""GPU: the block fed from HBM (lora_demod_core_work_device) makes the same calls and the same PDUs as the block
fed from host memory, for any chunking; and a wideband capture goes channeliser -> demod without leaving the GPU."""
import math

import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


def run_block(g, sf, f, ldr, feed, pieces):
    blk = g.demod(sf, ldr, cases.BETA, f)
    blk.enable_trace(1 << 16)
    pdus = []
    for p in pieces:
        pdus += blk.work(feed(p))
    tr = blk.trace().copy()
    st = blk.stats
    blk.close()
    return pdus, tr, st


@pytest.mark.parametrize("sf,f,ldr", [(7, 2, 0), (8, 2, 1), (9, 1, 0), (10, 4, 0), (12, 2, 0)])
def test_device_input_equals_host_input(sf, f, ldr, g):
    import torch
    n = 1 << sf
    cap, _, _ = cases.make_capture(sf, [20, 9, 33], seed=300 + sf, snr_db=10)
    cap = np.concatenate([cap, np.zeros(4 * n, np.complex64)])
    rng = np.random.default_rng(sf)
    cuts = [0] + sorted(set(int(v) for v in rng.integers(1, len(cap), 7))) + [len(cap)]
    # one chunk shorter than a symbol, one shorter than the history, the rest long
    cuts = sorted(set(cuts + [cuts[2] + n // 3, cuts[4] + 2 * n + 5]))
    cuts = [c for c in cuts if c <= len(cap)]
    pieces = [cap[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    want, tr_h, _ = run_block(g, sf, f, ldr, lambda p: p, pieces)
    got, tr_d, st = run_block(g, sf, f, ldr, lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda(), pieces)
    assert len(want) >= 2 and len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert np.array_equal(tr_d, tr_h)                 # same read indices, states, argmax and |X|^2 bits, call by call
    # and whole-capture feeding from the device against the oracle's block
    ref, _, _, _ = oracle.Demod(sf, bool(ldr), cases.BETA, f).stream(cap)
    whole, _, _ = run_block(g, sf, f, ldr, lambda p: torch.from_numpy(p).cuda(), [cap])
    assert len(whole) == len(ref)
    for a, b in zip(whole, ref):
        assert np.array_equal(a, b)


def test_wideband_capture_stays_on_the_gpu(g):
    """channeliser (device in, device out) -> demod block reading that output in place."""
    import torch
    sf, f, up = 8, 2, 4
    n = 1 << sf
    cap, _, _ = cases.make_capture(sf, [24, 40], seed=77, snr_db=None)
    spec = np.fft.fft(cap.astype(np.complex128)); m = len(cap)
    wide = np.zeros(m * up, np.complex128)
    wide[:m // 2] = spec[:m // 2]; wide[-(m - m // 2):] = spec[m // 2:]
    off = 0.17
    wb = np.concatenate([np.fft.ifft(wide) * up * np.exp(-2j * np.pi * off * np.arange(m * up)),
                         np.zeros(16 * n * up)]).astype(np.complex64)
    ch = g.channelizer(2 * math.pi * off, 1.0 / up)
    blk = g.demod(sf, False, cases.BETA, f)
    x = torch.from_numpy(wb).cuda()
    pdus = []
    for a in range(0, len(wb), 100_003):             # stream it through in pieces, everything device-resident
        pdus += blk.work(ch.work(x[a:a + 100_003]))
    ch.reset()
    host = g.demod(sf, False, cases.BETA, f).work(ch.work(wb))      # the same through host buffers
    assert len(pdus) == len(host) == 2
    for a, b in zip(pdus, host):
        assert np.array_equal(a, b)
    ch.close(); blk.close()


def test_wideband_capture_file(g, tmp_path):
    """A raw float32 I/Q file of a wideband capture through iqfile.demod_file with a channeliser: same PDUs as the
    arrays through host buffers, for a chunk size that cuts packets."""
    from importlib import import_module
    iqfile = import_module("gr-lora_amd.iqfile")
    sf, f, up, off = 7, 2, 8, -0.31
    n = 1 << sf
    cap, _, _ = cases.make_capture(sf, [16, 30, 11], seed=91, snr_db=None)
    spec = np.fft.fft(cap.astype(np.complex128)); m = len(cap)
    wide = np.zeros(m * up, np.complex128)
    wide[:m // 2] = spec[:m // 2]; wide[-(m - m // 2):] = spec[m // 2:]
    wb = np.concatenate([np.fft.ifft(wide) * up * np.exp(-2j * np.pi * off * np.arange(m * up)),
                         np.zeros(16 * n * up)]).astype(np.complex64)
    path = str(tmp_path / "wide.32iq")
    iqfile.write_iq(path, wb)
    ch = g.channelizer(2 * math.pi * off, 1.0 / up)
    want = g.demod(sf, False, cases.BETA, f).work(ch.work(wb))
    ch.reset()
    got = iqfile.demod_file(path, g.demod(sf, False, cases.BETA, f), chunk_samples=33_333, channelizer=ch)
    assert len(want) == 3 and len(got) == 3
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    ch.close()
