"""GPU: planned evaluation of the block (lora_demod_core_set_plan, include/lora_demod_core.h).

A plan only chooses WHICH windows are evaluated ahead of the state machine (all 16 lattice phases for preambles, every
SFD search the preamble rule can start, the header/payload windows a dry run predicts), never what a window's result
is: the trace -- read index, state, chirp offset, argmax and |X|^2 bits of every emulated general_work call
(demod_impl.cc:149-450) -- and the PDUs must be identical, byte for byte, to the on-demand block's, and equal the oracle's
to the L2 bar.  Also the multi-run launch the plan is built on (lora_hip_demod_batch_runs) against one strided launch
per run."""
import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


def run_block(g, sf, f, ldr, pieces, plan, feed=lambda p: p):
    blk = g.demod(sf, bool(ldr), cases.BETA, f, plan=plan)
    blk.enable_trace(1 << 17)
    pdus = []
    for p in pieces:
        pdus += blk.work(feed(p))
    tr = blk.trace().copy()
    st = blk.stats
    blk.close()
    return pdus, tr, st


def same_pdus(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def dense_capture(sf, n_packets, seed, snr_db=8, max_gap=12):
    rng = np.random.default_rng(seed)
    n = 1 << sf
    parts = [np.zeros(int(rng.integers(0, 3 * n)), np.complex64)]
    for pk in range(n_packets):
        parts.append(oracle.modulate(sf, oracle.random_symbols(sf, int(rng.integers(9, 40)), seed * 1000 + pk), cases.SYNC_WORD))
        parts.append(np.zeros(int(rng.integers(0, max_gap * n)), np.complex64))
    parts.append(np.zeros(8 * n, np.complex64))
    cap = np.concatenate(parts)
    if snr_db is None:
        return cap
    return oracle.awgn(cap, 10.0 ** (-snr_db / 10.0), seed * 7919 + 17, cases.squelch_gain(sf, 2, snr_db))


@pytest.mark.parametrize("host_iq", [True, False])
def test_batch_runs_equals_strided_launches(g, host_iq):
    import torch
    sf = 9; n = 1 << sf
    rng = np.random.default_rng(3)
    iq = (rng.standard_normal(200 * n) + 1j * rng.standard_normal(200 * n)).astype(np.complex64)
    h = g.capi.HipDemod(sf, 2, cases.BETA)
    runs = [(0, n, 37, 0, 0, g.capi.WIN_UP), (5 * n + 32, n // 16, 161, 32, n // 16, g.capi.WIN_UP),
            (17, n, 1, 500, 0, g.capi.WIN_UP), (3 * n, 0, 0, 0, 0, g.capi.WIN_UP), (100 * n + 7, 3, 99, 11, 5, g.capi.WIN_UP)]
    x = iq if host_iq else torch.from_numpy(iq).cuda()
    for kind in (g.capi.WIN_UP, g.capi.WIN_SFD):
        rk = [r[:5] + (kind,) for r in runs]
        idx, val = h.batch_runs(x, rk)
        k = 0
        for pos0, stride, count, off, step, _ in rk:
            if count == 0:
                continue
            i1, v1 = h.batch_strided(x, pos0, stride, count, chirp_off=off, chirp_off_step=step, kind=kind)
            assert np.array_equal(idx[k:k + count], i1)
            assert val[k:k + count].tobytes() == v1.tobytes()
            k += count
        assert k == len(idx)
    # a run whose last window leaves the capture is refused as a whole, like a host descriptor
    with pytest.raises(g.capi.LoraHipError):
        h.batch_runs(x, [(0, n, 4, 0, 0, 0), (198 * n, n, 3, 0, 0, 0)])
    h.close()


@pytest.mark.parametrize("name", sorted(cases.GOLDEN_CASES))
def test_planned_block_equals_on_demand_block_golden(name, g):
    p = cases.GOLDEN_CASES[name]
    cap, _, _ = cases.case_capture(p)
    a, tr_a, st_a = run_block(g, p["sf"], p["fft_factor"], p["ldr"], [cap], plan=0)
    b, tr_b, st_b = run_block(g, p["sf"], p["fft_factor"], p["ldr"], [cap], plan=2)
    assert st_a["plans"] == 0 and st_b["plans"] >= 1
    assert same_pdus(a, b)
    assert tr_a.tobytes() == tr_b.tobytes()


@pytest.mark.parametrize("sf,f,ldr,snr", [(7, 2, 0, 8), (8, 2, 1, 0), (8, 1, 0, None), (9, 4, 0, 5), (10, 2, 0, -3), (6, 2, 0, 10)])
def test_planned_block_many_packets(sf, f, ldr, snr, g):
    """Dense traffic, random gaps (some packets too close for the walker to be back in time): same trace and PDUs as the
    on-demand block, equal to the oracle's PDUs, in a fixed number of launches."""
    cap = dense_capture(sf, 40, seed=sf * 10 + f, snr_db=snr)
    a, tr_a, st_a = run_block(g, sf, f, ldr, [cap], plan=0)
    b, tr_b, st_b = run_block(g, sf, f, ldr, [cap], plan=1)
    print(f"sf{sf} f{f}: on demand {st_a['launches']} launches; planned {st_b['launches']} launches, "
          f"{st_b['replans']} replans, {st_b['on_demand']} on demand, {len(b)} PDUs, windows {st_a['windows']} -> {st_b['windows']}")
    assert st_b["plans"] == 1
    assert same_pdus(a, b) and len(b) >= 20
    assert tr_a.tobytes() == tr_b.tobytes()
    ref, _, _, _ = oracle.Demod(sf, bool(ldr), cases.BETA, f).stream(cap)
    if same_pdus(ref, a):       # (a near-tie may legitimately part the product from the oracle: test_gpu_l2_parity's bar)
        assert same_pdus(ref, b)
    assert st_b["launches"] <= 8 + 3 * (st_b["replans"] + st_b["on_demand"])
    assert st_b["launches"] < st_a["launches"]


@pytest.mark.parametrize("sf,f", [(7, 2), (9, 2)])
def test_planned_block_chunked_and_device_input(sf, f, g):
    import torch
    n = 1 << sf
    cap = dense_capture(sf, 25, seed=900 + sf, snr_db=6)
    rng = np.random.default_rng(sf)
    cuts = sorted(set([0, len(cap)] + [int(v) for v in rng.integers(1, len(cap), 6)] + [n // 3, 5 * n + 1]))
    pieces = [cap[x:y] for x, y in zip(cuts[:-1], cuts[1:])]
    want, tr_w, _ = run_block(g, sf, f, 0, pieces, plan=0)
    for feed in (lambda p: p, lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda()):
        got, tr_g, st = run_block(g, sf, f, 0, pieces, plan=2, feed=feed)
        assert st["plans"] >= 3
        assert same_pdus(want, got) and len(got) >= 10
        assert tr_w.tobytes() == tr_g.tobytes()


@pytest.mark.parametrize("sf,f,snr", [(7, 2, 8), (7, 1, None), (8, 4, 3)])
def test_planned_block_in_overlapping_stretches(sf, f, snr, g, monkeypatch):
    """A long input is planned in stretches whose GPU stages run ahead of the state machine (pipeline_run: lattice sweep of
    stretch k+1 and SFD stage of stretch k queued while stretch k-1 is walked, header/payload reads on the second lane).
    The cut changes which launch evaluates a window, never a result: trace and PDUs equal the on-demand block's byte for
    byte -- host input, device-resident input, and a chunked feed whose calls span several stretches each."""
    import torch
    n = 1 << sf
    n_packets = 4 * 65536 // (16 * 45) + 60
    cap = dense_capture(sf, n_packets, seed=4200 + sf + f, snr_db=snr, max_gap=6)
    assert len(cap) // n > 4 * 4096
    monkeypatch.setenv("LORA_CORE_STRETCH", "32768")   # short stretches (2048 symbols; the default is 180 000 positions): eight of them here
    a, tr_a, st_a = run_block(g, sf, f, 0, [cap], plan=0)
    b, tr_b, st_b = run_block(g, sf, f, 0, [cap], plan=1)
    print(f"sf{sf} f{f}: {len(cap) // n} symbols, {len(a)} PDUs; on demand {st_a['launches']} launches; pipelined {st_b['plans']} stretches, "
          f"{st_b['launches']} launches, {st_b['replans']} replans, {st_b['on_demand']} on demand")
    assert st_b["plans"] >= 4
    assert same_pdus(a, b) and len(b) >= n_packets // 3
    assert tr_a.tobytes() == tr_b.tobytes()
    assert st_b["launches"] <= 8 * st_b["plans"] + 3 * (st_b["replans"] + st_b["on_demand"])
    c, tr_c, st_c = run_block(g, sf, f, 0, [cap], plan=1, feed=lambda p: torch.from_numpy(np.ascontiguousarray(p)).cuda())
    assert same_pdus(a, c) and tr_a.tobytes() == tr_c.tobytes() and st_c["plans"] >= 4
    cut = (len(cap) * 5 // 9) | 1
    d, tr_d, st_d = run_block(g, sf, f, 0, [cap[:cut], cap[cut:]], plan=1)
    assert same_pdus(a, d) and tr_a.tobytes() == tr_d.tobytes() and st_d["plans"] >= 4


def test_plan_uses_preloaded_lattice(g):
    """Results handed in through lora_demod_core_preload for all 16 phases replace the plan's own lattice sweep."""
    sf, f = 8, 2; n = 1 << sf; step = n // 16
    cap = dense_capture(sf, 20, seed=77, snr_db=10)
    want, tr_w, st_w = run_block(g, sf, f, 0, [cap], plan=2)
    framed = cases.gr_framed(cap, sf)
    h = g.capi.HipDemod(sf, f, cases.BETA)
    blk = g.demod(sf, False, cases.BETA, f, plan=2)
    blk.enable_trace(1 << 17)
    for ph in range(16):
        cnt = (len(framed) - ph * step - n) // n + 1
        idx, val = h.batch_strided(framed, ph * step, n, cnt)
        blk.preload(ph * step, idx, val)
    got = blk.work(cap)
    st = blk.stats
    assert same_pdus(want, got) and blk.trace().tobytes() == tr_w.tobytes()
    # the lattice windows that the block can call are all there, so it launches only its SFD and READ stages
    print("preloaded:", st, "own sweep:", st_w)
    assert st["windows"] < st_w["windows"]
    blk.close(); h.close()


@pytest.mark.parametrize("fault_at", [0, 7, 333, 600])
def test_first_pass_misprediction_rolls_back(fault_at, g, monkeypatch):
    """The first pass of a plan predicts the squelch of header/payload windows from the lattice sweep; a prediction the
    exact result contradicts must put the state machine back before that packet and leave PDUs and trace as they are
    without a plan.  LORA_CORE_FAULT makes the n-th prediction wrong on purpose."""
    sf, f = 8, 2
    cap = dense_capture(sf, 30, seed=4242, snr_db=9)
    want, tr_w, _ = run_block(g, sf, f, 0, [cap], plan=0)
    monkeypatch.setenv("LORA_CORE_FAULT", str(fault_at))
    got, tr_g, st = run_block(g, sf, f, 0, [cap], plan=2)
    monkeypatch.delenv("LORA_CORE_FAULT")
    print(f"fault at prediction {fault_at}:", st)
    # (a fault that lands on a window that is squelched anyway, or behind the last S_RESET call, changes nothing)
    assert st["rollbacks"] == 1 if fault_at < 400 else st["rollbacks"] <= 1
    assert same_pdus(want, got) and len(got) >= 20
    assert tr_w.tobytes() == tr_g.tobytes()


def test_pinned_results_and_bins_only(g):
    """LORA_HIP_OUT_PINNED (results straight into the caller's page-locked arrays) and LORA_HIP_NO_VAL (bins only) give the
    same bits as the plain host-output call."""
    import ctypes as C
    capi = g.capi
    L = capi.lib()
    sf = 8; n = 1 << sf; count = 300
    rng = np.random.default_rng(11)
    iq = (rng.standard_normal(count * n + 5) + 1j * rng.standard_normal(count * n + 5)).astype(np.complex64)
    h = capi.HipDemod(sf, 2, cases.BETA)
    idx0, val0 = h.batch_strided(iq, 3, n, count, chirp_off=17)
    pi, pv = C.c_void_p(), C.c_void_p()
    assert L.lora_hip_host_alloc(C.byref(pi), count * 2) == 0 and L.lora_hip_host_alloc(C.byref(pv), count * 4) == 0
    try:
        idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_uint16)), (count,))
        val = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_float)), (count,))
        OUT_PINNED, NO_VAL = capi.OUT_PINNED, capi.NO_VAL
        idx[:] = 0xFFFF; val[:] = -7
        rc = L.lora_hip_demod_batch_strided(h._h, iq.ctypes.data, len(iq), 3, n, count, 17, 0, capi.WIN_UP, pi, pv, OUT_PINNED)
        assert rc == 0 and np.array_equal(idx, idx0) and val.tobytes() == val0.tobytes()
        idx[:] = 0xFFFF; val[:] = -7
        rc = L.lora_hip_demod_batch_strided(h._h, iq.ctypes.data, len(iq), 3, n, count, 17, 0, capi.WIN_UP, pi, None, OUT_PINNED | NO_VAL)
        assert rc == 0 and np.array_equal(idx, idx0) and np.all(val == -7)
        plain = np.full(count, 0xFFFF, np.uint16)
        rc = L.lora_hip_demod_batch_strided(h._h, iq.ctypes.data, len(iq), 3, n, count, 17, 0, capi.WIN_UP, plain.ctypes.data, None, NO_VAL)
        assert rc == 0 and np.array_equal(plain, idx0)
        # without the flag a missing value array is an error, as before
        assert L.lora_hip_demod_batch_strided(h._h, iq.ctypes.data, len(iq), 3, n, count, 17, 0, capi.WIN_UP, plain.ctypes.data, None, 0) == capi.EINVAL
    finally:
        L.lora_hip_host_free(pi); L.lora_hip_host_free(pv)
        h.close()


def test_dense_stream_in_short_chunks_switches_to_plans(g):
    """Default mode plans an input of at least 1024 symbols -- or of 192 when the stream has carried a packet per 160 symbols or more
    lately (demod_core.cpp: kPlanMinDense).  A dense stream fed in 256-symbol chunks: the first chunk is evaluated on demand, later
    ones are planned; a sparse one (a packet per ~400 symbols) never is.  Either way the trace is the on-demand block's."""
    sf, f = 7, 2
    n = 1 << sf
    cap = dense_capture(sf, 40, seed=77, max_gap=6)
    chunk = 256 * n
    pieces = [cap[k:k + chunk] for k in range(0, len(cap), chunk)]
    assert len(pieces) >= 6
    want, tr_w, st_w = run_block(g, sf, f, 0, pieces, plan=0)
    got, tr_g, st_g = run_block(g, sf, f, 0, pieces, plan=1)
    assert st_w["plans"] == 0 and st_g["plans"] >= len(pieces) - 3, (st_w, st_g)
    assert same_pdus(want, got) and len(got) >= 20 and tr_w.tobytes() == tr_g.tobytes()
    assert st_g["launches"] < st_w["launches"]
    sparse = dense_capture(sf, 6, seed=78, max_gap=700)
    pieces = [sparse[k:k + chunk] for k in range(0, len(sparse), chunk)]
    got, tr_g, st_g = run_block(g, sf, f, 0, pieces, plan=1)
    want, tr_w, st_w = run_block(g, sf, f, 0, pieces, plan=0)
    assert st_g["plans"] == 0 and same_pdus(want, got) and tr_w.tobytes() == tr_g.tobytes()
