"""One rank of the sharded paths on the product's block, started N times by tools/run_shard.py (or a test).

Every rank is a fresh process: gloo process group on the host (the data path has no collective), its own GPU
(LOCAL_RANK % device count, so that N ranks can be rehearsed on one GPU), its own demod block.
  cfg3  BASELINE configs[3]: independent SF10 fft_factor-2 captures, one packet each, round-robin-free contiguous shares
        (shard.partition), demodulated start to end by one rank, PDU lists gathered on rank 0 (shard.gather_pdus);
  cfg4  BASELINE configs[4]: ONE long SF9 fft_factor-2 capture with Poisson packet arrivals, cut into N time segments, every
        rank walks its own segment plus an overlap and holds nothing else of the capture (shard.RangeSource), rank 0
        chains the trajectories (shard.demod_long_capture_sharded).
Rank 0 then runs the same inputs through ONE sequential block and prints one JSON line: per-rank seconds, symbol times
per second, and whether the sharded PDU lists are identical to the sequential block's.
Captures come from the product-side generator (gr-lora_amd/synth.py); nothing under oracle/ is used."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

BETA = 25.0


def packet_symbols(sf, n_syms, seed):
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 1 << sf, n_syms)
    s[:8] &= ~3
    return s


def poisson_layout(sf, n_samples, mean_gap_s, fs, seed, force_across=None):
    """[(start, n_syms, seed)] of the packets of the long capture; force_across: a sample index some packet must straddle."""
    rng = np.random.default_rng(seed)
    n = 1 << sf
    out, at, k = [], 0, 0
    while True:
        gap = int(rng.exponential(mean_gap_s) * fs)
        ns = int(rng.integers(16, 129))
        flen = int((4 + 12.25 + ns + 4) * n) + 128
        start = at + gap
        if force_across is not None and start <= force_across and start + flen + gap > force_across and not (start < force_across < start + flen):
            start = max(at, force_across - flen // 2)          # move this packet over the cut
        if start + flen + 4 * n > n_samples:
            break
        out.append((start, ns, seed * 7919 + k))
        at = start + flen; k += 1
    return out


def make_loader(sf, layout, synth):
    n = 1 << sf
    starts = np.array([p[0] for p in layout], np.int64)

    def load(lo, hi):
        x = np.zeros(hi - lo, np.complex64)
        k = max(0, int(np.searchsorted(starts, lo)) - 1)
        while k < len(layout) and layout[k][0] < hi:
            st, ns, sd = layout[k]
            f = synth.modulate(sf, packet_symbols(sf, ns, sd))
            a, b = max(st, lo), min(st + len(f), hi)
            if b > a:
                x[a - lo:b - lo] = f[a - st:b - st]
            k += 1
        return x
    return load


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--captures", type=int, default=1024)
    ap.add_argument("--capture-symbols", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=3600.0)
    ap.add_argument("--overlap-symbols", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=4, help="blocks in flight per rank for cfg3")
    ap.add_argument("--skip-sequential", action="store_true", help="timing only: no comparison with one sequential block")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))

    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")       # host-side control plane only: barriers and gather_object
    import gr_lora_amd as g
    synth = importlib.import_module("gr-lora_amd.synth")
    shard = importlib.import_module("gr-lora_amd.shard")
    ndev = torch.cuda.device_count()
    dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)

    def barrier():
        if world > 1:
            dist.barrier()

    def gather(x):
        if world == 1:
            return [x]
        out = [None] * world if rank == 0 else None
        dist.gather_object(x, out, dst=0)
        return out

    res = {"n_ranks": world, "devices_visible": ndev}
    lists = lambda pdus: [np.asarray(p).tolist() for p in pdus]

    # ---- cfg3: independent captures --------------------------------------------------------------------------------
    sf, f = 10, 2
    n = 1 << sf

    def load_capture(cid):
        fr = synth.modulate(sf, packet_symbols(sf, args.capture_symbols, 4000 + cid))
        return np.concatenate([fr, np.zeros(4 * n, np.complex64)])

    lo, hi = shard.my_range(args.captures, rank, world)
    caps = {cid: load_capture(cid) for cid in range(lo, hi)}        # in host memory before the clock starts (as a file cache would be)
    make_block = lambda: g.demod(sf, False, BETA, f, device=dev)
    make_block().close()                                              # first use of the device outside the timed region
    barrier()
    t0 = time.perf_counter()
    local = shard.demod_captures(range(lo, hi), lambda cid: caps[cid], make_block, threads=args.threads)
    t_rank = time.perf_counter() - t0
    out = shard.gather_pdus(local, args.captures)
    t_all = gather(t_rank)
    cap_len = len(next(iter(caps.values()))) if caps else 0
    if rank == 0:
        r3 = {"captures": args.captures, "samples_per_capture": cap_len, "per_rank_s": [round(t, 4) for t in t_all],
              "pdus": sum(len(p) for p in out),
              "msym_per_s": round(args.captures * cap_len / n / max(t_all) / 1e6, 4)}
        if not args.skip_sequential:
            blk = make_block()
            t0 = time.perf_counter()
            seq = []
            for cid in range(args.captures):
                seq.append(lists(blk.work(caps[cid] if cid in caps else load_capture(cid)))); blk.reset()
            r3["sequential_s"] = round(time.perf_counter() - t0, 4)
            blk.close()
            r3["identical_to_sequential_block"] = [lists(p) for p in out] == seq
        res["cfg3"] = r3
    del caps
    barrier()

    # ---- cfg4: one long capture, time segments ------------------------------------------------------------------------
    sf, f, fs = 9, 2, 125e3
    n = 1 << sf
    n_samples = int(args.seconds * fs)
    cuts = shard.segment_cuts(n_samples, sf, max(world, 2))
    across = cuts[1] - (3 * n - 1) if len(cuts) > 2 else None         # capture index of the first cut: a packet straddles it
    layout = poisson_layout(sf, n_samples, 1.0, fs, 5, force_across=across)
    src = shard.RangeSource(n_samples, make_loader(sf, layout, synth))
    make_block = lambda: g.demod(sf, False, BETA, f, device=dev)
    walker = shard.BlockWalker(src, sf, make_block)
    walker.segments = world
    barrier()
    t0 = time.perf_counter()
    got = shard.demod_long_capture_sharded(src, walker, sf, overlap_symbols=args.overlap_symbols)
    t_rank = time.perf_counter() - t0
    walker.close()
    t_all = gather(t_rank)
    held = gather(src.loaded)
    spanned = gather(src.spanned)
    if rank == 0:
        pdus, info = got
        r4 = {"samples": n_samples, "packets_sent": len(layout), "pdus": len(pdus), "per_rank_s": [round(t, 4) for t in t_all],
              "samples_read_per_rank": held, "samples_spanned_per_rank": spanned, "fallbacks": info["fallbacks"], "trajectories": info["trajectories"],
              "msym_per_s": round(n_samples / n / max(t_all) / 1e6, 4),
              "note": "per_rank_s includes synthesising the segment's samples on the host (the loader) and, on rank 0, the chaining"}
        if not args.skip_sequential:
            blk = make_block()
            whole = src[0:n_samples]
            t0 = time.perf_counter()
            seq = blk.work(whole)
            r4["sequential_s"] = round(time.perf_counter() - t0, 4)
            blk.close()
            r4["identical_to_sequential_block"] = lists(pdus) == lists(seq)
            if across is not None:
                r4["packet_straddles_first_cut"] = any(st < across < st + int((12.25 + ns) * n) for st, ns, _ in layout)
        res["cfg4"] = r4
        print(json.dumps(res), flush=True)
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
