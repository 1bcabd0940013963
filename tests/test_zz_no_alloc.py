"""The entry point that allocates nothing (include/lora_hip_demod.h: lora_hip_demod_batch_strided_noalloc).

Its ticket counters live in a file-scope __device__ array of the library instead of a buffer of the handle.
Checked through the C ABI against lora_hip_demod_batch_strided on the same device buffers, bit for bit (integer
indices and the float maxima: same kernels, same launch geometry), on two calls in a row, the second with a
smaller input -- the scratch keeps the first call's counters, so a missing reset would show there.
Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


def _run_both(g, h, t_iq, pos0, stride, count, **kw):
    import torch
    dev = t_iq.device
    oi = torch.full((count,), -1, dtype=torch.int16, device=dev); ov = torch.full((count,), -7.0, dtype=torch.float32, device=dev)
    ri = torch.empty(count, dtype=torch.int16, device=dev); rv = torch.empty(count, dtype=torch.float32, device=dev)
    h.batch_strided(t_iq, pos0, stride, count, out_idx=ri, out_val=rv, **kw)      # the existing entry point
    h.batch_strided_noalloc(t_iq, pos0, stride, count, oi, ov, **kw)              # only enqueues
    torch.cuda.synchronize()
    a, b = oi.cpu().numpy().view(np.uint16), ov.cpu().numpy()
    c, d = ri.cpu().numpy().view(np.uint16), rv.cpu().numpy()
    assert np.array_equal(a, c)
    assert np.array_equal(b.view(np.uint32), d.view(np.uint32))
    return a, b


@pytest.mark.parametrize("sf,f", [(12, 2), (8, 2), (10, 4), (7, 1)])
def test_noalloc_equals_batch_strided_twice(sf, f, g):
    import torch
    n = 1 << sf
    dev = torch.device("cuda:0")
    cap, _, _ = cases.make_capture(sf, [24, 17], seed=900 + sf, snr_db=3)
    big = np.tile(cap, -(-(3000 * n) // len(cap)))[:3000 * n] if sf <= 10 else np.tile(cap, 8)
    t_big = torch.from_numpy(big).to(dev)
    h = g.HipDemod(sf, f, cases.BETA)
    stream = torch.cuda.current_stream()
    h.set_stream(stream.cuda_stream)
    count = len(big) // n
    i1, v1 = _run_both(g, h, t_big, 0, n, count)                              # first call: many window groups
    assert np.all(i1 < n * f) and np.all(v1 >= 0)
    small = big[n // 4: n // 4 + 5 * n + 3]
    t_small = torch.from_numpy(np.ascontiguousarray(small)).to(dev)
    _run_both(g, h, t_small, 1, n, 5, chirp_off=n // 4)                       # second call: smaller input, other offset
    # the SFD sweep (stride N/16, stepping chirp offset) goes through the same entry point
    _run_both(g, h, t_small, 0, n // 16, 16 * 4, chirp_off=0, chirp_off_step=n // 16, kind=g.capi.WIN_SFD)
    h.set_stream(None)
    h.close()


def test_noalloc_two_handles_share_the_scratch_in_stream_order(g):
    """Two handles (different SF) on one stream: every call finds the scratch as the previous one left it."""
    import torch
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    hs = []
    for sf, f in [(9, 2), (11, 1)]:
        h = g.HipDemod(sf, f, cases.BETA); h.set_stream(stream.cuda_stream); hs.append((sf, h))
    rng = np.random.default_rng(5)
    for rep in range(3):
        for sf, h in hs:
            n = 1 << sf
            count = int(rng.integers(1, 700))
            x = (rng.standard_normal(count * n) + 1j * rng.standard_normal(count * n)).astype(np.complex64)
            _run_both(g, h, torch.from_numpy(x).to(dev), 0, n, count)
    for _, h in hs:
        h.set_stream(None); h.close()


def test_noalloc_two_threads_two_handles_two_streams(g):
    """Two blocks on two scheduler threads: each handle owns a slot of the per-device scratch, so calls that overlap in
    time on one GPU (two streams, two threads) no longer share counters.  Every result is compared, bit for bit, with
    the allocating entry point's; the device's free memory does not move across the no-allocation calls."""
    import threading
    import torch
    dev = torch.device("cuda:0")
    sf, f = 10, 2
    n = 1 << sf
    count = 6000                                    # ~30 window groups per workgroup: the launches overlap for a while
    rng = np.random.default_rng(11)
    work = []
    for t in range(2):
        x = (rng.standard_normal(count * n) + 1j * rng.standard_normal(count * n)).astype(np.complex64)
        h = g.HipDemod(sf, f, cases.BETA)
        st = torch.cuda.Stream(device=dev)
        h.set_stream(st.cuda_stream)
        t_iq = torch.from_numpy(x).to(dev)
        ri = torch.empty(count, dtype=torch.int16, device=dev); rv = torch.empty(count, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        h.batch_strided(t_iq, 0, n, count, out_idx=ri, out_val=rv)            # reference results (synchronous)
        outs = [(torch.empty(count, dtype=torch.int16, device=dev), torch.empty(count, dtype=torch.float32, device=dev)) for _ in range(12)]
        work.append((h, st, t_iq, ri, rv, outs))
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    errs = []
    start = threading.Barrier(2)

    def run(k):
        h, st, t_iq, ri, rv, outs = work[k]
        try:
            start.wait()
            for oi, ov in outs:                       # back to back, only enqueued: both threads' kernels are in flight together
                h.batch_strided_noalloc(t_iq, 0, n, count, oi, ov)
            st.synchronize()
        except Exception as e:                        # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert torch.cuda.mem_get_info(dev)[0] == free0   # nothing was allocated or freed on the device
    for h, st, t_iq, ri, rv, outs in work:
        a, b = ri.cpu().numpy().view(np.uint16), rv.cpu().numpy().view(np.uint32)
        for oi, ov in outs:
            assert np.array_equal(oi.cpu().numpy().view(np.uint16), a)
            assert np.array_equal(ov.cpu().numpy().view(np.uint32), b)
        h.set_stream(None); h.close()


@pytest.mark.parametrize("fail_on", ["alloc", "noalloc"])
def test_failed_launch_then_alternating_entry_points(fail_on, g, monkeypatch):
    """A launch that fails leaves its ticket counters undefined (LORA_HIP_FAULT: the n-th launch of the handle reports a failure
    and fills the counters it was about to use with garbage).  The handle's own buffer and its scratch slot each carry a flag of
    their own, so whichever entry point runs next clears exactly the buffer it uses, and alternating them afterwards gives the
    results of an undisturbed handle."""
    import torch
    dev = torch.device("cuda:0")
    sf, f = 9, 2
    n = 1 << sf
    count = 1500
    rng = np.random.default_rng(21)
    x = (rng.standard_normal(count * n) + 1j * rng.standard_normal(count * n)).astype(np.complex64)
    t_iq = torch.from_numpy(x).to(dev)
    good = g.HipDemod(sf, f, cases.BETA)
    ref_i, ref_v = good.batch_strided(t_iq, 0, n, count)
    good.close()
    monkeypatch.setenv("LORA_HIP_FAULT", "2")
    h = g.HipDemod(sf, f, cases.BETA)
    monkeypatch.delenv("LORA_HIP_FAULT")
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)

    def alloc():
        i, v = h.batch_strided(t_iq, 0, n, count)
        assert np.array_equal(i, ref_i) and np.array_equal(v.view(np.uint32), ref_v.view(np.uint32))

    def noalloc():
        oi.fill_(-1); ov.fill_(-7.0)
        h.batch_strided_noalloc(t_iq, 0, n, count, oi, ov)
        torch.cuda.synchronize()
        assert np.array_equal(oi.cpu().numpy().view(np.uint16), ref_i)
        assert np.array_equal(ov.cpu().numpy().view(np.uint32), ref_v.view(np.uint32))

    first, second = (alloc, noalloc) if fail_on == "alloc" else (noalloc, alloc)
    first()                                           # launch 1
    with pytest.raises(RuntimeError):                 # launch 2 fails; garbage in the counters `first` uses
        first()
    for _ in range(2):                                # the other buffer first, then the one that was left undefined
        second()
        first()
    h.set_stream(None); h.close()


def test_noalloc_slots_run_out_and_come_back(g):
    """LORA_HIP_NOALLOC_MAX_HANDLES handles per GPU own a scratch slot; one more still works everywhere else and is refused
    here with a clear error; a slot given back at destroy is handed to the next handle."""
    import torch
    dev = torch.device("cuda:0")
    L = g.capi.lib()
    n = 128
    t = torch.zeros(4 * n, dtype=torch.complex64, device=dev)
    oi = torch.empty(4, dtype=torch.int16, device=dev); ov = torch.empty(4, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    hs = [g.HipDemod(7, 1, cases.BETA) for _ in range(g.capi.NOALLOC_MAX_HANDLES)]
    call = lambda h: L.lora_hip_demod_batch_strided_noalloc(h._h, t.data_ptr(), 4 * n, 0, n, 4, 0, 0, 0, oi.data_ptr(), ov.data_ptr())
    assert all(call(h) == g.capi.OK for h in hs)
    extra = g.HipDemod(7, 1, cases.BETA)
    assert call(extra) == g.capi.ENOMEM
    assert b"scratch slots" in L.lora_hip_last_hip_error(extra._h)
    i0, v0 = extra.batch_strided(t, 0, n, 4)                                  # the allocating entry point does not need a slot
    assert len(i0) == 4
    extra.close()
    hs.pop().close()
    again = g.HipDemod(7, 1, cases.BETA)
    assert call(again) == g.capi.OK
    torch.cuda.synchronize()
    for h in hs + [again]:
        h.close()


def test_noalloc_rejects_bad_sizes(g):
    import torch
    dev = torch.device("cuda:0")
    h = g.HipDemod(8, 2, cases.BETA)
    n = 256
    t = torch.zeros(4 * n, dtype=torch.complex64, device=dev)
    oi = torch.empty(8, dtype=torch.int16, device=dev); ov = torch.empty(8, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()            # the handle runs on a stream of its own here
    L = g.capi.lib()
    call = lambda n_samples, pos0, stride, count: L.lora_hip_demod_batch_strided_noalloc(
        h._h, t.data_ptr(), n_samples, pos0, stride, count, 0, 0, 0, oi.data_ptr(), ov.data_ptr())
    assert call(4 * n, 0, n, (1 << 30) + 1) == g.capi.EINVAL                  # beyond the stated maximum
    assert b"exceed the maximum" in L.lora_hip_last_hip_error(h._h)
    assert call(4 * n, 0, n, 5) == g.capi.EINVAL                               # last window past the buffer
    assert call(4 * n, 3 * n + 1, n, 1) == g.capi.EINVAL
    assert call(n - 1, 0, n, 1) == g.capi.EINVAL
    assert call(4 * n, 0, n, 0) == g.capi.OK                                   # nothing to do
    assert L.lora_hip_demod_batch_strided_noalloc(h._h, None, 4 * n, 0, n, 1, 0, 0, 0, oi.data_ptr(), ov.data_ptr()) == g.capi.EINVAL
    assert call(4 * n, 0, n, 4) == g.capi.OK
    torch.cuda.synchronize()
    h.close()
