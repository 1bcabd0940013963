"""The two lanes of a handle through the C ABI (include/lora_hip_demod.h: LORA_HIP_ASYNC with page-locked results,
LORA_HIP_LANE1, lora_hip_demod_sync / _mark / _wait): what the planned block queues ahead of its state machine
(DESIGN.md section 4).  Results must be the bits of the synchronous calls whichever lane ran them and however many calls
were queued before anybody waited.  No counterpart in the reference (its general_work is synchronous)."""
import ctypes as C

import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


class Pinned:
    """page-locked (idx, val) arrays from lora_hip_host_alloc"""
    def __init__(self, L, n):
        self.L, self.pi, self.pv = L, C.c_void_p(), C.c_void_p()
        assert L.lora_hip_host_alloc(C.byref(self.pi), n * 2) == 0 and L.lora_hip_host_alloc(C.byref(self.pv), n * 4) == 0
        self.idx = np.ctypeslib.as_array(C.cast(self.pi, C.POINTER(C.c_uint16)), (n,))
        self.val = np.ctypeslib.as_array(C.cast(self.pv, C.POINTER(C.c_float)), (n,))
        self.idx[:] = 0xFFFF; self.val[:] = -7

    def free(self):
        self.L.lora_hip_host_free(self.pi); self.L.lora_hip_host_free(self.pv)


def test_queued_sweeps_marks_and_the_second_lane(g):
    capi = g.capi
    L = capi.lib()
    sf = 9; n = 1 << sf; F = 2
    rng = np.random.default_rng(21)
    n_win = 6000
    iq = (rng.standard_normal(n_win * n + 64) + 1j * rng.standard_normal(n_win * n + 64)).astype(np.complex64)
    h = capi.HipDemod(sf, F, cases.BETA)
    h.stage_iq(iq)
    # three sweeps, and what the synchronous calls give for them
    sweeps = [(5, n, n_win - 1, 0, 0, capi.WIN_UP),                    # long, register-resident taps
              (n // 16, n // 16, 16 * 300, 3, n // 16, capi.WIN_SFD),  # an SFD search: stepping chirp offset
              (7 * n + 9, n, 200, 131, 0, capi.WIN_UP)]                # a packet's header/payload windows
    ref = [h.batch_strided(None, p0, st, cnt, chirp_off=off, chirp_off_step=step, kind=kind) for p0, st, cnt, off, step, kind in sweeps]
    bufs = [Pinned(L, s[2]) for s in sweeps]
    A = capi.ASYNC | capi.OUT_PINNED | capi.IQ_STAGED
    try:
        def queue(k, extra):
            p0, st, cnt, off, step, kind = sweeps[k]
            return L.lora_hip_demod_batch_strided(h._h, None, len(iq), p0, st, cnt, off, step, kind, bufs[k].pi, bufs[k].pv, A | extra)
        for rep in range(3):                                            # the handle's staging is reused from call to call
            for b in bufs:
                b.idx[:] = 0xFFFF; b.val[:] = -7
            assert queue(0, 0) == 0 and L.lora_hip_demod_mark(h._h, 0, 0) == 0
            assert queue(1, 0) == 0 and L.lora_hip_demod_mark(h._h, 0, 1) == 0
            assert queue(2, capi.LANE1) == 0                            # does not wait behind the two sweeps
            assert L.lora_hip_demod_sync(h._h, 1) == 0
            assert np.array_equal(bufs[2].idx, ref[2][0]) and bufs[2].val.tobytes() == ref[2][1].tobytes()
            assert L.lora_hip_demod_wait(h._h, 0) == 0
            assert np.array_equal(bufs[0].idx, ref[0][0]) and bufs[0].val.tobytes() == ref[0][1].tobytes()
            assert L.lora_hip_demod_wait(h._h, 1) == 0
            assert np.array_equal(bufs[1].idx, ref[1][0]) and bufs[1].val.tobytes() == ref[1][1].tobytes()
            assert L.lora_hip_demod_sync(h._h, 0) == 0
        # LORA_HIP_OUT_DIRECT: the kernel stores into the page-locked arrays itself; same bits, synchronous and queued, either lane
        for extra in (0, capi.LANE1):
            for b in bufs:
                b.idx[:] = 0xFFFF; b.val[:] = -7
            p0, st, cnt, off, step, kind = sweeps[2]
            assert L.lora_hip_demod_batch_strided(h._h, None, len(iq), p0, st, cnt, off, step, kind, bufs[2].pi, bufs[2].pv,
                                                  capi.OUT_PINNED | capi.IQ_STAGED | capi.OUT_DIRECT | extra) == 0
            assert np.array_equal(bufs[2].idx, ref[2][0]) and bufs[2].val.tobytes() == ref[2][1].tobytes()
            assert queue(1, capi.OUT_DIRECT | extra) == 0 and queue(0, capi.OUT_DIRECT | extra) == 0
            assert L.lora_hip_demod_sync(h._h, 1 if extra else 0) == 0
            for k in (0, 1):
                assert np.array_equal(bufs[k].idx, ref[k][0]) and bufs[k].val.tobytes() == ref[k][1].tobytes()
        # a mark that was never set counts as reached; bad lanes and marks are refused
        assert L.lora_hip_demod_wait(h._h, 7) == 0
        assert L.lora_hip_demod_wait(h._h, 8) == capi.EINVAL and L.lora_hip_demod_mark(h._h, 0, -1) == capi.EINVAL
        assert L.lora_hip_demod_mark(h._h, 2, 0) == capi.EINVAL and L.lora_hip_demod_sync(h._h, 2) == capi.EINVAL
        # no synchronisation needs buffers the copies can run into behind the call's return: pageable results are refused
        plain_i, plain_v = np.zeros(200, np.uint16), np.zeros(200, np.float32)
        p0, st, cnt, off, step, kind = sweeps[2]
        assert L.lora_hip_demod_batch_strided(h._h, None, len(iq), p0, st, cnt, off, step, kind, plain_i.ctypes.data, plain_v.ctypes.data,
                                              capi.ASYNC | capi.IQ_STAGED) == capi.EINVAL
        # the second lane has no staging copy of its own: host samples are refused there
        assert L.lora_hip_demod_batch_strided(h._h, iq.ctypes.data, len(iq), p0, st, cnt, off, step, kind, plain_i.ctypes.data,
                                              plain_v.ctypes.data, capi.LANE1) == capi.EINVAL
    finally:
        L.lora_hip_demod_sync(h._h, 0); L.lora_hip_demod_sync(h._h, 1)
        for b in bufs:
            b.free()
        h.close()


def test_run_lists_queued_back_to_back_on_both_lanes(g):
    """lora_hip_demod_batch_runs copies its run list to HBM from page-locked staging of the handle behind the call's return
    (two blocks per lane, used in turn): four calls in a row on a lane, none waited for, must not overwrite each other's list."""
    capi = g.capi
    L = capi.lib()
    sf = 8; n = 1 << sf
    rng = np.random.default_rng(22)
    iq = (rng.standard_normal(3000 * n) + 1j * rng.standard_normal(3000 * n)).astype(np.complex64)
    h = capi.HipDemod(sf, 2, cases.BETA)
    h.stage_iq(iq)
    lists = []
    for k in range(8):
        runs = []
        for _ in range(3 + 5 * k):
            cnt = int(rng.integers(1, 40))
            stride = int(rng.choice([n, n // 16]))
            p0 = int(rng.integers(0, len(iq) - n - cnt * stride))
            runs.append((p0, stride, cnt, int(rng.integers(0, n)), n // 16 if stride != n else 0, capi.WIN_SFD if stride != n else capi.WIN_UP))
        lists.append(runs)
    ref = [h.batch_runs(None, runs) for runs in lists]
    bufs = [Pinned(L, len(r[0])) for r in ref]
    try:
        for k, runs in enumerate(lists):
            r = np.zeros(len(runs), capi.WIN_RUN_DTYPE)
            for j, u in enumerate(runs):
                r["pos0"][j], r["stride"][j], r["count"][j], r["chirp_off"][j], r["chirp_off_step"][j], r["kind"][j] = u
            lane = capi.LANE1 if k % 2 else 0
            rc = L.lora_hip_demod_batch_runs(h._h, None, len(iq), r.ctypes.data, len(r), bufs[k].pi, bufs[k].pv,
                                             capi.ASYNC | capi.OUT_PINNED | capi.IQ_STAGED | lane)
            assert rc == 0
            r[:] = 0                                                     # the caller's list is its own again when the call returns
        assert L.lora_hip_demod_sync(h._h, 0) == 0 and L.lora_hip_demod_sync(h._h, 1) == 0
        for k in range(len(lists)):
            assert np.array_equal(bufs[k].idx, ref[k][0]) and bufs[k].val.tobytes() == ref[k][1].tobytes(), k
    finally:
        L.lora_hip_demod_sync(h._h, 0); L.lora_hip_demod_sync(h._h, 1)
        for b in bufs:
            b.free()
        h.close()
