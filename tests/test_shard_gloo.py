"""Multi-process (gloo, world_size 2, CPU) test of the N>1 path: captures are partitioned over
ranks with no data-path collective, each rank demodulates its own, rank 0 gathers the PDU lists
in capture order.  The per-rank block is the oracle here (no GPU in this test); the sharding and
gather code under test is the product's (gr-lora_amd/shard.py)."""
import os
import socket
import sys

import numpy as np
import pytest

import cases
import oracle

N_CAPTURES = 5
SF, F = 7, 2


def _capture(cid):
    cap, _, _ = cases.make_capture(SF, [10 + cid, 8], seed=900 + cid)
    return cap


class _OracleBlock:
    def __init__(self):
        self.d = oracle.Demod(SF, False, cases.BETA, F)

    def work(self, samples):
        return self.d.stream(samples)[0]


def _worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    shard = importlib.import_module("gr-lora_amd.shard")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    lo, hi = shard.my_range(N_CAPTURES, rank, world)
    local = shard.demod_captures(range(lo, hi), _capture, _OracleBlock)
    out = shard.gather_pdus(local, N_CAPTURES, dst=0)
    if rank == 0:
        q.put([[p.tolist() for p in c] for c in out])
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    import importlib
    shard = importlib.import_module("gr-lora_amd.shard")
    for n in (0, 1, 7, 8, 1024):
        for w in (1, 2, 3, 8):
            parts = shard.partition(n, w)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharding_equals_sequential():
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
    ref = [[p.tolist() for p in _OracleBlock().work(_capture(c))] for c in range(N_CAPTURES)]
    assert got == ref
    assert sum(len(c) for c in ref) >= N_CAPTURES       # every capture produced PDUs


# ---- channeliser: time segments of one wideband capture over 2 ranks (shard.channelize_segment) -------------------
class _OracleChannelizer:
    """Stand-in with the product's interface (taps_per_filter, seek, work), built on the oracle.  The oracle has no
    closed-form seek: it reaches the stream state at n_consumed by running the reference recurrence over that many
    inputs (schedule(), which touches no samples) -- exactly what the product's seek must be equivalent to -- and
    then evaluates its own positions and weights over (history ++ segment) with sequential float sums."""

    def __init__(self, rate):
        self.rate = rate
        self.r = oracle.PfbArbResampler(rate)
        self.taps_per_filter = self.r.taps_per_filter

    def seek(self, n_consumed, history):
        self.r = oracle.PfbArbResampler(self.rate)
        if n_consumed:
            self.r.schedule(n_consumed)
        assert len(history) == self.taps_per_filter - 1
        self._hist = np.ascontiguousarray(history, np.complex64)

    def work(self, x):
        K, f32 = self.taps_per_filter, np.float32
        pos, acc = self.r.schedule(len(x))                     # this segment's (32*i_in + arm, d_acc) per output
        t = np.zeros(32 * K, f32); t[:len(self.r.taps)] = self.r.taps
        d = np.zeros(32 * K, f32); d[:len(self.r.taps) - 1] = np.diff(self.r.taps)
        arms, darms = t.reshape(K, 32).T[:, ::-1], d.reshape(K, 32).T[:, ::-1]     # reversed, as fir_filter stores them
        s = np.concatenate([self._hist, np.ascontiguousarray(x, np.complex64)])
        out = np.empty(len(pos), np.complex64)
        for m, (p, a) in enumerate(zip(pos, acc)):
            i, j = int(p) // 32, int(p) % 32
            seg = s[i:i + K]
            re0 = im0 = re1 = im1 = f32(0)
            for k in range(K):                                 # sequential float sums, products rounded first
                re0 = f32(re0 + f32(seg[k].real * arms[j, k])); im0 = f32(im0 + f32(seg[k].imag * arms[j, k]))
                re1 = f32(re1 + f32(seg[k].real * darms[j, k])); im1 = f32(im1 + f32(seg[k].imag * darms[j, k]))
            out[m] = complex(f32(re0 + f32(re1 * a)), f32(im0 + f32(im1 * a)))
        return out


def _chan_worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    shard = importlib.import_module("gr-lora_amd.shard")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(1501) + 1j * rng.standard_normal(1501)).astype(np.complex64)
    y = shard.channelize_segment(x, _OracleChannelizer(0.7), rank, world)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(y, gathered, dst=0)                     # host gather of outputs, no samples exchanged
    if rank == 0:
        q.put(np.concatenate(gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_channeliser_segments_equal_sequential():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(1501) + 1j * rng.standard_normal(1501)).astype(np.complex64)
    want = oracle.PfbArbResampler(0.7).work(x)
    assert len(got) == len(want) and np.array_equal(got, want)
