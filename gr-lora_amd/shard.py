"""Multi-GPU sharding of the demod path: one process per GPU, no data-path collective.

Two ways the path shards, both without any exchange of samples or spectra:

(i)  independent captures: a capture (one channelised stream, one gr::lora::demod instance in the
     reference's flowgraph) is demodulated start to end by exactly one rank
     (partition / demod_captures / gather_pdus);
(ii) time segments of one long capture (design A): the reference's window grid depends on history
     (every SFD lock shifts it by ol*N/16 + 5N/4, lib/demod_impl.cc:344), so a segment cannot be
     demodulated cold.  But outside READ_HEADER / READ_PAYLOAD a call's result depends only on its
     read index, and read indices stay on the N/16 lattice: every rank sweeps the stride-N windows
     of its own time segment for the 16 lattice phases (sharded_detect_sweep), rank 0 gathers the
     (idx, val) pairs (6 bytes per window), hands them to the block (demod.preload) and replays the
     reference's state machine over the whole capture: same grid, same PDUs as the sequential run.

Ranks only meet for host-side gathers (torch.distributed gather_object): RCCL/xGMI carry nothing.
"""
import numpy as np


def partition(n_items, world_size):
    """Contiguous, balanced [start, stop) ranges: rank r owns items partition(...)[r]."""
    base, extra = divmod(n_items, world_size)
    out, start = [], 0
    for r in range(world_size):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


def my_range(n_items, rank, world_size):
    return partition(n_items, world_size)[rank]


def _demod_some(capture_ids, load_capture, make_block):
    out, blk = {}, None
    for cid in capture_ids:
        if blk is None:
            blk = make_block()
        out[cid] = [np.asarray(p, np.uint16) for p in blk.work(load_capture(cid))]
        if hasattr(blk, "reset"):
            blk.reset()                 # same block, fresh stream: keeps the tables and GPU buffers
        else:
            close = getattr(blk, "close", None)
            if close:
                close()
            blk = None
    if blk is not None and hasattr(blk, "close"):
        blk.close()
    return out


def demod_captures(capture_ids, load_capture, make_block, threads=1):
    """Demodulate the captures of this rank.  load_capture(id) -> complex64 array;
    make_block() -> an object with work(samples) -> list of u16 arrays (gr-lora_amd.demod or any
    stand-in with the same interface).  Returns {id: [pdu, ...]}.

    threads > 1: that many blocks (one handle and one HIP stream each) work through the captures side by side, as
    the reference's flowgraph runs one demod block per channel on a scheduler thread of its own.  A short capture is
    bound by the latency of its few launches and of its host-to-device copy, not by GPU time, so captures in flight
    overlap those; the C ABI is re-entrant per handle and releases the interpreter lock while it works."""
    ids = list(capture_ids)
    if threads <= 1 or len(ids) <= 1:
        return _demod_some(ids, load_capture, make_block)
    from concurrent.futures import ThreadPoolExecutor
    threads = min(threads, len(ids))
    shares = [ids[t::threads] for t in range(threads)]
    out = {}
    with ThreadPoolExecutor(threads) as pool:
        for part in pool.map(lambda share: _demod_some(share, load_capture, make_block), shares):
            out.update(part)
    return out


def gather_pdus(local, n_items, dst=0):
    """Host gather of per-capture PDU lists on rank `dst`, ordered by capture id.
    Works with any initialised torch.distributed backend (nccl = RCCL on the GPU box, gloo in
    the CPU tests); returns None on the other ranks.  Without a process group: single rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local[i] for i in range(n_items)]
    world, rank = dist.get_world_size(), dist.get_rank()
    payload = {int(k): [p.tolist() for p in v] for k, v in local.items()}
    gathered = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for part in gathered:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"capture {k} was demodulated by two ranks")
            merged[k] = [np.asarray(p, np.uint16) for p in v]
    missing = [i for i in range(n_items) if i not in merged]
    if missing:
        raise RuntimeError(f"captures {missing[:8]} were demodulated by no rank")
    return [merged[i] for i in range(n_items)]


# ---- (ii) time segments of one long capture ---------------------------------------------------

LATTICE = 16     # OVERLAP_FACTOR, lib/demod_impl.cc:36: read indices are multiples of N/16


def framed_slice(samples, sf, lo, hi):
    """Samples [lo, hi) of the framed stream = 3N-1 zeros (set_history(3N), lib/demod_impl.cc:103)
    followed by the capture."""
    hist = 3 * (1 << sf) - 1
    out = np.zeros(hi - lo, np.complex64)
    s_lo, s_hi = max(lo - hist, 0), min(hi - hist, len(samples))
    if s_hi > s_lo:
        out[s_lo + hist - lo:s_hi + hist - lo] = samples[s_lo:s_hi]
    return out


def lattice_partition(n_samples, sf, world_size):
    """[q0, q1) ranges of lattice indices per rank; window q starts at framed-stream index q*N/16."""
    n, step = 1 << sf, (1 << sf) // LATTICE
    n_stream = 3 * n - 1 + n_samples
    n_starts = 0 if n_stream < n else (n_stream - n) // step + 1
    return partition(n_starts, world_size)


def sweep_segment(sweep, samples, sf, q0, q1):
    """Stride-N up-path sweeps of the lattice indices [q0, q1), one per lattice phase.
    sweep(segment, pos0, count) -> (idx u16[count], val f32[count]) evaluates the windows starting
    at segment[pos0 + k*N] (HipSweeper, or any stand-in).  Returns [(start, idx, val)] x 16 with
    start = framed-stream index of the first window of that phase (None if the phase is empty)."""
    n, step = 1 << sf, (1 << sf) // LATTICE
    out = []
    if q1 <= q0:
        return [(None, np.zeros(0, np.uint16), np.zeros(0, np.float32))] * LATTICE
    seg_lo, seg_hi = q0 * step, (q1 - 1) * step + n
    seg = framed_slice(samples, sf, seg_lo, seg_hi)
    for phase in range(LATTICE):
        qp = q0 + ((phase - q0) % LATTICE)
        if qp >= q1:
            out.append((None, np.zeros(0, np.uint16), np.zeros(0, np.float32)))
            continue
        count = (q1 - 1 - qp) // LATTICE + 1
        idx, val = sweep(seg, qp * step - seg_lo, count)
        out.append((qp * step, np.asarray(idx, np.uint16), np.asarray(val, np.float32)))
    return out


def merge_sweeps(parts, sf):
    """parts[r] = sweep_segment(...) of rank r, in rank order -> one (start, idx, val) per phase.
    Checks that the pieces of a phase are contiguous on its stride-N lattice."""
    n = 1 << sf
    merged = []
    for phase in range(LATTICE):
        start, idxs, vals, nxt = None, [], [], None
        for part in parts:
            s, i, v = part[phase]
            if s is None:
                continue
            if start is None:
                start = s
            elif s != nxt:
                raise RuntimeError(f"phase {phase}: segment starts at {s}, expected {nxt}")
            idxs.append(i); vals.append(v)
            nxt = s + len(i) * n
        if start is not None:
            merged.append((start, np.concatenate(idxs), np.concatenate(vals)))
    return merged


class HipSweeper:
    """sweep() on the GPU: stages the segment once, then one strided launch per lattice phase."""

    def __init__(self, sf, fft_factor, beta=25.0, device=0):
        from . import capi
        self._capi = capi
        self.h = capi.HipDemod(sf, fft_factor, beta, device=device)
        self.n = 1 << sf
        self._seg = None

    def __call__(self, seg, pos0, count):
        if self._seg is not seg:
            self.h.stage_iq(seg)
            self._seg = seg
        return self.h.batch_strided(None, pos0, self.n, count, 0, 0, self._capi.WIN_UP)

    def close(self):
        self.h.close()


def sharded_detect_sweep(samples, sf, sweep, dst=0):
    """Every rank sweeps its own time segment; rank `dst` returns the merged per-phase results
    [(start, idx, val)], the others None.  Without a process group: one rank does it all."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if multi else (1, 0)
    q0, q1 = lattice_partition(len(samples), sf, world)[rank]
    mine = sweep_segment(sweep, samples, sf, q0, q1)
    if not multi:
        return merge_sweeps([mine], sf)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(mine, gathered, dst=dst)
    return merge_sweeps(gathered, sf) if rank == dst else None


def demod_long_capture(samples, make_block, sf, sweep, dst=0):
    """Design A end to end: sharded detect sweep, then rank `dst` replays the state machine.
    make_block() -> gr-lora_amd.demod (needs preload() and work()).  Returns the PDU list on rank
    `dst`, None elsewhere."""
    runs = sharded_detect_sweep(samples, sf, sweep, dst)
    if runs is None:
        return None
    blk = make_block()
    for start, idx, val in runs:
        blk.preload(start, idx, val)
    pdus = blk.work(samples)
    blk._shard_stats = dict(getattr(blk, "stats", {}))
    return pdus


# ---- channeliser: time segments of one wideband capture (include/lora_hip_chan.h, lora_hip_chan_seek) ----------
def channelize_segment(samples, chan, rank, world):
    """This rank's share of channelising one long capture: the contiguous segment my_range(len(samples), rank, world)
    of the input, entered with lora_hip_chan_seek (closed-form state, taps_per_filter - 1 samples of history read
    from the capture).  The per-rank outputs, concatenated in rank order, are the sequential stream's output;
    no samples are exchanged between ranks."""
    lo, hi = my_range(len(samples), rank, world)
    k1 = chan.taps_per_filter - 1
    hist = np.zeros(k1, np.complex64)
    take = min(k1, lo)
    if take:
        hist[k1 - take:] = samples[lo - take:lo]
    chan.seek(lo, hist)
    return chan.work(np.ascontiguousarray(samples[lo:hi]))
