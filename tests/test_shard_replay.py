"""Sharded replay of one long capture (gr-lora_amd/shard.py section ii-b): every rank demodulates its own time
segment, entered with lora_demod_core_seek on each lattice phase, and rank 0 chains the segments where two trajectories
are idle together.

CPU part: the segmentation, the joining and the chaining are the product's code; the block is the oracle's
(lo_demod_stream_framed = the block entered in S_RESET at a framed-stream index), since there is no GPU here.  The
chained PDU list must equal the sequential block's for any number of segments, with packets that straddle the cuts,
and when a segment has to be walked again; a 2-rank gloo run must give rank 0 the same list.
GPU part: the same with the product's block (seek, work, idle_log)."""
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
N = 1 << SF


def capture(seed=31, n_packets=14, sf=SF, max_gap=40, snr_db=None):
    rng = np.random.default_rng(seed)
    n = 1 << sf
    parts, spans, at = [np.zeros(int(rng.integers(0, 3 * n)), np.complex64)], [], 0
    at = len(parts[0])
    for pk in range(n_packets):
        frame = oracle.modulate(sf, oracle.random_symbols(sf, int(rng.integers(9, 40)), seed * 1000 + pk), cases.SYNC_WORD)
        spans.append((at + 4 * n, at + len(frame) - 4 * n))       # the frame carries 4N zeros in front and behind
        parts.append(frame); at += len(frame)
        gap = np.zeros(int(rng.integers(0, max_gap * n)), np.complex64)
        parts.append(gap); at += len(gap)
    parts.append(np.zeros(8 * n, np.complex64))
    cap = np.concatenate(parts)
    if snr_db is not None:
        cap = oracle.awgn(cap, 10.0 ** (-snr_db / 10.0), seed * 7919 + 17, cases.squelch_gain(sf, 2, snr_db))
    return cap, spans


def idle_from_trace(tr, n):
    """What lora_demod_core_idle_log reports, from the oracle's per-call trace."""
    runs = []
    out_before = np.concatenate([[0], np.cumsum(tr["state"] == oracle.S_OUT)])[:-1]
    for i in np.flatnonzero(tr["state"] == oracle.S_DETECT_PREAMBLE):
        p = int(tr["pos"][i])
        if runs and runs[-1][1] + n == p:
            runs[-1][1] = p
        else:
            runs.append([p, p, int(out_before[i])])
    return np.array(runs, np.uint64).reshape(-1, 3)


class OracleWalker:
    def __init__(self, samples, sf=SF, f=F, segments=1):
        self.samples, self.sf, self.d, self.segments = samples, sf, oracle.Demod(sf, False, cases.BETA, f), segments
        self.walks = 0

    def __call__(self, a, b):
        pdus, tr = self.d.stream_framed(shard.framed_slice(self.samples, self.sf, a, b), a)
        self.walks += 1
        return dict(pdus=pdus, idle=idle_from_trace(tr, 1 << self.sf), end=b)


def lists(pdus):
    return [np.asarray(p).tolist() for p in pdus]


def test_seeked_walker_from_zero_is_the_sequential_block():
    cap, _ = capture()
    ref, tr, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap, want_trace=True)
    w = OracleWalker(cap)
    t = w(0, 3 * N - 1 + len(cap))
    assert lists(t["pdus"]) == lists(ref) and len(ref) >= 10
    assert np.array_equal(t["idle"], idle_from_trace(tr, N))


@pytest.mark.parametrize("world", [2, 3, 5, 8])
@pytest.mark.parametrize("seed,snr", [(31, None), (32, 6), (33, None)])
def test_chained_segments_equal_the_sequential_block(world, seed, snr):
    cap, spans = capture(seed, snr_db=snr)
    ref, _, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap)
    w = OracleWalker(cap, segments=world)
    pdus, info = shard.demod_long_capture_sharded(cap, w, SF, overlap_symbols=160)
    assert lists(pdus) == lists(ref) and len(ref) >= 10
    assert info["segments"] == world and info["fallbacks"] == 0
    assert info["trajectories"] == [1] + [16] * (world - 1)


def test_a_packet_straddles_a_cut():
    """The cut of a 2-segment run is put into the payload of a packet on purpose."""
    cap, spans = capture(41)
    ref, _, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap)
    hist = 3 * N - 1
    lo, hi = spans[len(spans) // 2]
    cut = N * ((hist + (lo + hi) // 2) // N)                      # framed index inside that packet
    end = hist + len(cap)
    cuts = [0, cut, end]
    w = OracleWalker(cap)
    trajs = [shard.walk_segment(w, SF, cuts[k], cuts[k + 1], end, 160, first=k == 0) for k in range(2)]
    pdus, fallbacks = shard.chain_segments(trajs, cuts, SF, w)
    assert lists(pdus) == lists(ref) and fallbacks == 0
    # the segment that starts inside the packet did not produce that packet: the chain took it from the first one
    joined = [t.get("joined_at") for t in trajs[1][1:]]
    assert all(j is not None for j in joined)


def test_segment_that_never_joins_is_walked_again():
    """With no overlap the chain is over before the next segment is idle: every segment is then walked again as the
    continuation of the chain, and the result is still the sequential block's."""
    cap, _ = capture(35, n_packets=8)
    ref, _, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap)
    w = OracleWalker(cap, segments=3)
    pdus, info = shard.demod_long_capture_sharded(cap, w, SF, overlap_symbols=0)
    assert lists(pdus) == lists(ref)
    assert info["fallbacks"] >= 1


def test_join_and_splice_bookkeeping():
    n = 8
    a = dict(pdus=["a0", "a1", "a2"], idle=np.array([[0, 40, 0], [120, 200, 1], [304, 344, 3]], np.uint64), end=400)
    b = dict(pdus=["b0", "b1"], idle=np.array([[131, 171, 0], [176, 232, 0], [320, 400, 1]], np.uint64), end=480)
    # 131 is off a's lattice (120 + 8k); 176 = 120 + 7*8 is on it
    assert shard._join(a, b, n, 0) == (176, 1, 1)
    assert shard._join(a, b, n, 190) == (192, 1, 1)
    assert shard._join(a, b, n, 210) == (320, 2, 2)
    t = shard._splice(a, b, n, 0)
    assert t["pdus"] == ["a0", "b0", "b1"] and t["joined_at"] == 176 and t["end"] == 480
    assert t["idle"].tolist() == [[0, 40, 0], [120, 168, 1], [176, 232, 1], [320, 400, 2]]
    assert shard._join(a, dict(pdus=[], idle=np.zeros((0, 3), np.uint64), end=0), n, 0) is None


def _worker(rank, world, port, q):
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    sh = importlib.import_module("gr-lora_amd.shard")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cap, _ = capture(41)
    out = sh.demod_long_capture_sharded(cap, OracleWalker(cap), SF, overlap_symbols=160, dst=0)
    if rank == 0:
        q.put((lists(out[0]), out[1]))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_replay_equals_sequential_block():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, info = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cap, _ = capture(41)
    ref, _, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap)
    assert got == lists(ref) and len(ref) >= 10
    assert info["segments"] == 2 and info["fallbacks"] == 0


# ---- GPU: the product's block as the walker ----------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("sf,f,world,plan", [(7, 2, 2, 2), (8, 2, 4, 1), (9, 2, 3, 0), (8, 4, 8, 2), (10, 1, 2, 1)])
def test_gpu_sharded_replay_equals_sequential(sf, f, world, plan):
    import gr_lora_amd as g
    cap, _ = capture(500 + sf, n_packets=12, sf=sf, snr_db=8)
    seq = g.demod(sf, False, cases.BETA, f)
    ref = lists(seq.work(cap))
    seq.close()
    assert len(ref) >= 5
    for on_device in (False, True):
        w = shard.BlockWalker(cap, sf, lambda: g.demod(sf, False, cases.BETA, f, plan=plan), on_device=on_device)
        w.segments = world
        pdus, info = shard.demod_long_capture_sharded(cap, w, sf, overlap_symbols=192)
        w.close()
        assert lists(pdus) == ref
        assert info["fallbacks"] == 0


@pytest.mark.gpu
def test_gpu_seek_and_idle_log_match_the_oracle():
    """lora_demod_core_seek / idle_log against the oracle's framed-stream entry, from an index in the middle of a packet."""
    import gr_lora_amd as g
    cap, spans = capture(61, n_packets=6)
    hist = 3 * N - 1
    a = (N // 16) * ((hist + (spans[2][0] + spans[2][1]) // 2) // (N // 16))
    b = hist + len(cap)
    want = OracleWalker(cap)(a, b)
    w = shard.BlockWalker(cap, SF, lambda: g.demod(SF, False, cases.BETA, F, plan=2))
    got = w(a, b)
    w.close()
    assert lists(got["pdus"]) == lists(want["pdus"]) and len(want["pdus"]) >= 3
    assert np.array_equal(got["idle"], want["idle"])


@pytest.mark.gpu
@pytest.mark.parametrize("plan", [1, 2])
@pytest.mark.parametrize("chunk_symbols", [37, 1500, 4000])
def test_gpu_idle_log_with_chunked_feed_matches_the_oracle(plan, chunk_symbols):
    """Multi-call streaming with a plan: a work() call that begins while the block is already in S_DETECT_PREAMBLE runs its
    first pass on top of the open idle run; undoing that pass (a long idle chunk with no S_RESET call, a rollback) must
    give the run back as it was.  The log stays sorted, disjoint and equal to the oracle's for any chunking."""
    import gr_lora_amd as g
    cap, _ = capture(77, n_packets=8, max_gap=400, snr_db=9)        # long idle stretches between packets
    ref, tr, _, _ = oracle.Demod(SF, False, cases.BETA, F).stream(cap, want_trace=True)
    want = idle_from_trace(tr, N)
    blk = g.demod(SF, False, cases.BETA, F, plan=plan)
    pdus = []
    step = chunk_symbols * N + 13
    for a in range(0, len(cap), step):
        pdus += blk.work(cap[a:a + step])
        log = blk.idle_log()
        assert np.all(log[1:, 0] > log[:-1, 1]) and np.all(log[:, 1] >= log[:, 0])      # sorted and disjoint after every call
    got = blk.idle_log().copy()
    blk.close()
    assert lists(pdus) == lists(ref) and len(ref) >= 5
    assert np.array_equal(got, want)
