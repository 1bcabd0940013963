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


class RangeSource:
    """A capture that is read by range: load(lo, hi) -> complex64 samples [lo, hi) of the capture (a file reader, a
    generator, a network fetch).  Everything in this module only takes len() and contiguous slices of `samples`, so a
    rank that is handed a RangeSource holds its own time segment plus the overlap and nothing else (a numpy array or
    a numpy.memmap works the same way)."""

    def __init__(self, n_samples, load):
        self.n_samples, self.load = int(n_samples), load
        self.loaded = 0                    # samples read so far (ranges that are walked again count again)
        self.lo_min, self.hi_max = None, None     # the part of the capture this process ever touched

    def __len__(self):
        return self.n_samples

    @property
    def spanned(self):
        """Samples between the first and the last one this process ever read."""
        return 0 if self.lo_min is None else self.hi_max - self.lo_min

    def __getitem__(self, sl):
        if not isinstance(sl, slice) or sl.step not in (None, 1):
            raise TypeError("RangeSource is read in contiguous slices only")
        lo, hi, _ = sl.indices(self.n_samples)
        hi = max(hi, lo)
        self.loaded += hi - lo
        if hi > lo:
            self.lo_min = lo if self.lo_min is None else min(self.lo_min, lo)
            self.hi_max = hi if self.hi_max is None else max(self.hi_max, hi)
        x = np.ascontiguousarray(self.load(lo, hi), np.complex64)
        if len(x) != hi - lo:
            raise ValueError(f"load({lo}, {hi}) returned {len(x)} samples")
        return x


def framed_slice(samples, sf, lo, hi):
    """Samples [lo, hi) of the framed stream = 3N-1 zeros (set_history(3N), lib/demod_impl.cc:103)
    followed by the capture."""
    hist = 3 * (1 << sf) - 1
    if lo >= hist and hi - hist <= len(samples):          # wholly inside the capture: a view, nothing is copied
        return samples[lo - hist:hi - hist]
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


# ---- (ii-b) sharded replay: every rank demodulates its own time segment -----------------------------
#
# Design A above shards the 16-phase preamble sweep and leaves the state machine -- and with it the SFD searches, the
# header/payload windows and the whole capture's host-to-device copy -- to one rank.  Here the state machine is sharded
# too.  It rests on one property of the reference's block (lib/demod_impl.cc:214-273): at an S_RESET call everything
# but the read index is cleared, and at an S_DETECT_PREAMBLE call the argmax history is the four windows before the read
# index; so two blocks that make such a call at the same read index behave identically from there on, wherever each was
# started.  A rank therefore starts a block at its segment's first sample (lora_demod_core_seek) once per lattice phase
# the sequential block could be on, walks its segment plus an overlap, and logs when it is idle
# (lora_demod_core_idle_log).  The phases collapse onto one trajectory as soon as a packet has been locked (the SFD lock
# depends on the signal, not on the phase the search was entered on), so only a short prefix is walked 16 times.
# Rank 0 then chains the segments: the trajectory that is the sequential block's so far, continued through its overlap,
# is matched against the next segment's trajectories at the first read index where both are idle; PDUs before that
# index come from the former, PDUs after it from the latter.  A segment for which no such index exists is walked again
# from the last idle index of the chain (correct for any input; slower only then).  Nothing is exchanged between ranks
# but the trajectories (PDUs and idle runs).

def _join(ta, tb, n, min_pos):
    """First read index >= min_pos at which both trajectories make an S_DETECT_PREAMBLE call (idle runs on the same
    stride-N lattice that overlap) -> (pos, run index in ta, run index in tb) or None."""
    ia, ib = ta["idle"], tb["idle"]
    # runs are sorted and disjoint: skip those that end before min_pos
    a = int(np.searchsorted(ia[:, 1], min_pos)) if len(ia) else 0
    b = int(np.searchsorted(ib[:, 1], min_pos)) if len(ib) else 0
    best = None
    while a < len(ia) and b < len(ib):
        fa, la = int(ia[a][0]), int(ia[a][1]); fb, lb = int(ib[b][0]), int(ib[b][1])
        if (fa - fb) % n == 0:
            p = max(fa, fb)
            if p < min_pos:
                p += -(-(min_pos - p) // n) * n
            if p <= min(la, lb):
                best = (p, a, b)
                break
        if la < lb:
            a += 1
        else:
            b += 1
    return best


def _splice(head, tail, n, min_pos):
    """The trajectory that follows `head` until the two are idle together, `tail` from there on; None if they never are."""
    j = _join(head, tail, n, min_pos)
    if j is None:
        return None
    p, a, b = j
    keep = int(head["idle"][a][2])                    # PDUs head emitted before that idle run
    drop = int(tail["idle"][b][2])
    pdus = list(head["pdus"][:keep]) + list(tail["pdus"][drop:])
    idle = [tuple(int(v) for v in r) for r in head["idle"][:a]]
    if int(head["idle"][a][0]) < p:
        idle.append((int(head["idle"][a][0]), p - n, keep))
    idle.append((p, int(tail["idle"][b][1]), keep))
    idle += [(int(r[0]), int(r[1]), int(r[2]) - drop + keep) for r in tail["idle"][b + 1:]]
    return dict(pdus=pdus, idle=np.array(idle, np.uint64).reshape(-1, 3), end=tail["end"], joined_at=p)


def segment_cuts(n_samples, sf, world_size):
    """Framed-stream indices (multiples of N) at which the segments start, plus the end of the stream."""
    n = 1 << sf
    n_stream = 3 * n - 1 + n_samples
    syms = n_stream // n
    return [n * ((syms * r) // world_size) for r in range(world_size)] + [n_stream]


def walk_segment(walker, sf, lo, hi, end, overlap_symbols, first):
    """All trajectories of the segment [lo, hi) of the framed stream (end = its length).
    walker(a, b) -> dict(pdus, idle (k,3) uint64, end=b): a block entered in S_RESET at framed index a, fed [a, b).
    The first segment has one trajectory (the stream's true start); every other one has 16, one per lattice phase the
    sequential block may arrive on: phase 0 walks the whole segment plus the overlap, the others only until they are
    idle together with phase 0 (tried over the overlap first, then over twice as much, ...)."""
    n, step = 1 << sf, (1 << sf) // LATTICE
    stop = min(end, hi + overlap_symbols * n)
    main = walker(lo, stop)
    if first:
        return [main]
    out = [main]
    for phase in range(1, LATTICE):
        span = max(overlap_symbols, 64) * n
        while True:
            b = min(stop, lo + span)
            w = walker(lo + phase * step, b) if lo + phase * step + 4 * n <= b else dict(pdus=[], idle=np.zeros((0, 3), np.uint64), end=b)
            t = _splice(w, main, n, lo)
            if t is not None or b >= stop:
                out.append(t if t is not None else w)
                break
            span *= 2
    return out


def chain_segments(trajs, cuts, sf, walker=None):
    """trajs[k] = walk_segment(...) of segment k, in order -> the sequential block's PDU list.
    walker: used only for a segment none of whose trajectories is ever idle together with the chain so far."""
    n = 1 << sf
    cur = trajs[0][0]
    fallbacks = 0
    for k in range(1, len(trajs)):
        best = None
        for t in trajs[k]:
            j = _join(cur, t, n, cuts[k])
            if j is not None and (best is None or j[0] < best[0][0]):
                best = (j, t)
        if best is not None:
            cur = _splice(cur, best[1], n, cuts[k])
            continue
        # never idle together: walk this segment again as the continuation of the chain
        if walker is None:
            raise RuntimeError(f"segment {k}: no trajectory joins the chain and no walker was given")
        fallbacks += 1
        if len(cur["idle"]):
            a = len(cur["idle"]) - 1
            p = int(cur["idle"][a][1])                    # the chain's last idle call: a block in S_RESET 5 calls earlier is there too
            stop = trajs[k][0]["end"]
            if p - 5 * n >= 0:
                redo = walker(p - 5 * n, stop)
                cur = _splice(cur, redo, n, p)
                if cur is not None:
                    continue
        cur = walker(0, trajs[k][0]["end"])               # no idle call at all so far: the sequential run
    return cur["pdus"], fallbacks


class BlockWalker:
    """walker() on a demod block (gr-lora_amd.demod): seek, feed the framed samples, read PDUs and idle log.
    samples: a numpy array, a numpy.memmap or a RangeSource -- only the ranges a walk covers are ever read."""

    def __init__(self, samples, sf, make_block, on_device=False):
        self.samples, self.sf, self.blk, self.on_device = samples, sf, make_block(), on_device
        self.walks = 0

    def __call__(self, a, b):
        hist = 3 * (1 << self.sf) - 1
        if a == 0 and b > hist:            # the stream's true start: the block's own history zeros, no framed copy
            self.blk.reset()
            x = self.samples[:b - hist]
        else:
            self.blk.seek(a)
            x = framed_slice(self.samples, self.sf, a, b)
        if self.on_device:
            import torch
            x = torch.from_numpy(x).cuda()
        pdus = self.blk.work(x)
        self.walks += 1
        return dict(pdus=pdus, idle=self.blk.idle_log(), end=b)

    def close(self):
        self.blk.close()


def demod_long_capture_sharded(samples, walker, sf, overlap_symbols=1024, dst=0):
    """Sharded replay end to end: every rank walks its own time segment, rank `dst` chains them and returns
    (pdus, info); the others return None.  Without a process group one rank walks every segment in turn
    (world_size segments when `walker` carries a `segments` attribute, else one)."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(), dist.get_rank()) if multi else (int(getattr(walker, "segments", 1)), 0)
    cuts = segment_cuts(len(samples), sf, world)
    end = cuts[-1]
    if multi:
        mine = walk_segment(walker, sf, cuts[rank], cuts[rank + 1], end, overlap_symbols, first=rank == 0)
        gathered = [None] * world if rank == dst else None
        dist.gather_object(mine, gathered, dst=dst)
        if rank != dst:
            return None
    else:
        gathered = [walk_segment(walker, sf, cuts[k], cuts[k + 1], end, overlap_symbols, first=k == 0) for k in range(world)]
    pdus, fallbacks = chain_segments(gathered, cuts, sf, walker)
    return pdus, dict(segments=world, fallbacks=fallbacks, trajectories=[len(t) for t in gathered])


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
