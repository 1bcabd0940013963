"""Time-segment sharding of one long capture (SURVEY 8e design A, gr-lora_amd/shard.py).

CPU part: the segmentation, per-phase sweeps, gather and merge are the product's code; the window
evaluator is the oracle (no GPU here).  The merged 16-phase sweep must not depend on the number of
ranks, and a 2-rank gloo run must hand rank 0 exactly the single-rank result.
GPU part: demod.preload + work must give the very same calls and PDUs as the sequential block.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

import cases
import oracle

shard = importlib.import_module("gr-lora_amd.shard")
SF, F = 7, 2


def _capture():
    cap, _, _ = cases.make_capture(SF, [12, 9, 15], seed=4242)
    return cap


class _OracleSweep:
    """sweep(segment, pos0, count) on the CPU oracle"""

    def __init__(self, sf=SF, f=F):
        self.d = oracle.Demod(sf, False, cases.BETA, f)
        self.n = 1 << sf

    def __call__(self, seg, pos0, count):
        pos = pos0 + self.n * np.arange(count, dtype=np.uint64)
        idx, val, _ = self.d.window_argmax(seg, pos, 0, 0, want_margin=False)
        return idx, val


def _as_lists(runs):
    return [(int(s), i.tolist(), v.tolist()) for s, i, v in runs]


def test_framed_slice_is_history_zeros_then_capture():
    cap = (np.arange(1, 1001) + 1j).astype(np.complex64)
    h = 3 * (1 << SF) - 1
    full = shard.framed_slice(cap, SF, 0, h + len(cap))
    assert not full[:h].any() and np.array_equal(full[h:], cap)
    for lo, hi in ((0, 10), (h - 5, h + 5), (h + 100, h + 300), (h + 990, h + 1010)):
        ref = np.concatenate([full, np.zeros(64, np.complex64)])[lo:hi]
        assert np.array_equal(shard.framed_slice(cap, SF, lo, hi), ref)


def test_lattice_partition_covers_every_window_start():
    n = 1 << SF
    for n_samples in (0, 1, n, 5 * n + 3, 40 * n):
        n_stream = 3 * n - 1 + n_samples
        total = (n_stream - n) // (n // 16) + 1
        for w in (1, 2, 3, 8):
            parts = shard.lattice_partition(n_samples, SF, w)
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))


def test_merged_sweep_is_independent_of_the_number_of_segments():
    cap = _capture()
    sweep = _OracleSweep()
    one = shard.merge_sweeps([shard.sweep_segment(sweep, cap, SF, *shard.lattice_partition(len(cap), SF, 1)[0])], SF)
    assert len(one) == 16
    n = 1 << SF
    # phase p starts at p*N/16 and its windows tile the framed stream
    n_stream = 3 * n - 1 + len(cap)
    for p, (start, idx, val) in enumerate(one):
        assert start == p * n // 16 and len(idx) == len(val) == (n_stream - n - start) // n + 1
    # against direct evaluation of the framed stream
    framed = shard.framed_slice(cap, SF, 0, n_stream)
    for start, idx, val in one[::5]:
        ridx, rval = sweep(framed, start, len(idx))
        assert np.array_equal(idx, ridx) and np.array_equal(val, rval)
    for world in (2, 3, 5, 8):
        parts = [shard.sweep_segment(sweep, cap, SF, q0, q1) for q0, q1 in shard.lattice_partition(len(cap), SF, world)]
        assert _as_lists(shard.merge_sweeps(parts, SF)) == _as_lists(one)


def test_merge_rejects_a_gap():
    cap = _capture()
    sweep = _OracleSweep()
    (a0, a1), (b0, b1) = shard.lattice_partition(len(cap), SF, 2)
    parts = [shard.sweep_segment(sweep, cap, SF, a0, a1 - 16), shard.sweep_segment(sweep, cap, SF, b0, b1)]
    with pytest.raises(RuntimeError):
        shard.merge_sweeps(parts, SF)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    sh = importlib.import_module("gr-lora_amd.shard")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    runs = sh.sharded_detect_sweep(_capture(), SF, _OracleSweep(), dst=0)
    if rank == 0:
        q.put(_as_lists(runs))
    else:
        assert runs is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sweep_equals_single_rank():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == _as_lists(shard.sharded_detect_sweep(_capture(), SF, _OracleSweep()))


# ---- GPU: preload + replay == sequential block ------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("sf,f,world", [(7, 2, 1), (8, 2, 4), (9, 2, 3), (8, 4, 2), (10, 1, 2)])
def test_gpu_preloaded_replay_equals_sequential(sf, f, world):
    import gr_lora_amd as g
    cap, _, _ = cases.make_capture(sf, [20, 11, 33, 8, 16], seed=77 + sf)
    seq = g.demod(sf, False, cases.BETA, f)
    seq.enable_trace(len(cap) // (1 << sf) * 2 + 64)
    ref_pdus = seq.work(cap)
    ref_trace = seq.trace().copy()
    ref_stats = seq.stats
    seq.close()
    assert len(ref_pdus) >= 1

    sweep = shard.HipSweeper(sf, f, cases.BETA)
    parts = [shard.sweep_segment(sweep, cap, sf, q0, q1) for q0, q1 in shard.lattice_partition(len(cap), sf, world)]
    runs = shard.merge_sweeps(parts, sf)
    sweep.close()
    blk = g.demod(sf, False, cases.BETA, f)
    blk.enable_trace(len(ref_trace) + 64)
    for start, idx, val in runs:
        blk.preload(start, idx, val)
    pdus = blk.work(cap)
    tr = blk.trace()
    stats = blk.stats
    blk.close()
    assert len(tr) == len(ref_trace)
    for fld in ("pos", "state", "chirp_off", "idx", "consumed", "n_sfd"):
        assert np.array_equal(tr[fld], ref_trace[fld]), fld
    assert np.array_equal(tr["val"].view(np.uint32), ref_trace["val"].view(np.uint32))
    assert [p.tolist() for p in pdus] == [p.tolist() for p in ref_pdus]
    # the replay launches only SFD sweeps and header/payload windows
    assert stats["launches"] < ref_stats["launches"]


@pytest.mark.gpu
def test_gpu_demod_long_capture_single_process():
    import gr_lora_amd as g
    sf, f = 9, 2
    cap, _, _ = cases.make_capture(sf, [40, 25, 31], seed=5)
    seq = g.demod(sf, False, cases.BETA, f)
    ref = [p.tolist() for p in seq.work(cap)]
    seq.close()
    sweep = shard.HipSweeper(sf, f, cases.BETA)
    got = shard.demod_long_capture(cap, lambda: g.demod(sf, False, cases.BETA, f), sf, sweep)
    sweep.close()
    assert [p.tolist() for p in got] == ref and len(ref) == 3


@pytest.mark.gpu
def test_gpu_preload_argument_checks():
    import gr_lora_amd as g
    blk = g.demod(8, False, cases.BETA, 2)
    with pytest.raises(g.capi.LoraHipError):
        blk.preload(5, np.zeros(4, np.uint16), np.zeros(4, np.float32))      # not on the N/16 lattice
    blk.preload(16, np.zeros(0, np.uint16), np.zeros(0, np.float32))         # empty: accepted
    blk.close()


# ---- property-based: segmentation/merge for arbitrary sizes, with a position-only fake evaluator ----------

def test_merge_matches_direct_evaluation_for_random_sizes():
    from hypothesis import given, settings, strategies as st

    class FakeSweep:
        """(idx, val) depend only on the framed-stream position of a window: sum of its first two samples."""
        def __init__(self, n):
            self.n = n

        def __call__(self, seg, pos0, count):
            p = pos0 + self.n * np.arange(count)
            v = (seg[p].real + 2 * seg[p + self.n - 1].real).astype(np.float32)
            return (v.astype(np.int64) % 65536).astype(np.uint16), v

    @settings(max_examples=60, deadline=None)
    @given(sf=st.integers(6, 9), n_samples=st.integers(0, 6000), world=st.integers(1, 9))
    def check(sf, n_samples, world):
        n = 1 << sf
        cap = (np.arange(1, n_samples + 1) % 251).astype(np.complex64)
        sweep = FakeSweep(n)
        parts = [shard.sweep_segment(sweep, cap, sf, q0, q1) for q0, q1 in shard.lattice_partition(n_samples, sf, world)]
        runs = shard.merge_sweeps(parts, sf)
        framed = shard.framed_slice(cap, sf, 0, 3 * n - 1 + n_samples)
        seen = 0
        for start, idx, val in runs:
            assert start % (n // 16) == 0 and start < n
            ridx, rval = sweep(framed, start, len(idx))
            assert np.array_equal(idx, ridx) and np.array_equal(val, rval)
            seen += len(idx)
        # every lattice start whose window fits in the framed stream is covered exactly once
        total = (len(framed) - n) // (n // 16) + 1 if len(framed) >= n else 0
        assert seen == total

    check()
