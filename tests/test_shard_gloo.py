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
