"""Multi-GPU sharding of the demod path: one process per GPU, independent captures per rank.

The path shards without any data-path exchange: a capture (one channelised stream,
one gr::lora::demod instance in the reference's flowgraph) is demodulated start to end by exactly
one rank; ranks only meet to gather the PDU lists on rank 0, in capture order.  That gather is
host-side control traffic (torch.distributed gather_object), not a device collective: RCCL/xGMI
carry nothing.
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


def demod_captures(capture_ids, load_capture, make_block):
    """Demodulate the captures of this rank.  load_capture(id) -> complex64 array;
    make_block() -> an object with work(samples) -> list of u16 arrays (gr-lora_amd.demod or any
    stand-in with the same interface).  Returns {id: [pdu, ...]}."""
    out = {}
    for cid in capture_ids:
        blk = make_block()
        out[cid] = [np.asarray(p, np.uint16) for p in blk.work(load_capture(cid))]
        close = getattr(blk, "close", None)
        if close:
            close()
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
