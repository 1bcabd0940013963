"""The sharded paths as N FRESH PROCESSES on the product's block, in small.

The CPU tests of the sharding logic (tests/test_shard_gloo.py, test_shard_replay.py) run the oracle's block under gloo;
the other GPU tests simulate ranks inside one process.  Here 2 rank processes (tools/shard_rank.py: gloo on
127.0.0.1, both on cuda:0 of this box, a demod block each) run
  (i)  BASELINE configs[3] in small: 12 SF10 captures through shard.demod_captures + gather_pdus, and
  (ii) BASELINE configs[4] in small: one 40 s SF9 capture with Poisson arrivals and a packet that straddles the cut, through
       shard.demod_long_capture_sharded with every rank reading only its own segment + overlap (shard.RangeSource),
and rank 0 compares both PDU lists with ONE sequential GPU block's.  This test only starts the processes and reads the line."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("ranks", [2, 3])
def test_rank_processes_equal_the_sequential_block(ranks):
    import run_shard
    rcs, out = run_shard.run(ranks, ["--captures", "12", "--capture-symbols", "40", "--seconds", "40", "--overlap-symbols", "192",
                                     "--threads", "2"], timeout=300)
    assert all(rc == 0 for rc in rcs), (rcs, out[-2000:])
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_ranks"] == ranks
    c3, c4 = line["cfg3"], line["cfg4"]
    assert c3["identical_to_sequential_block"] and c3["pdus"] >= 12 and len(c3["per_rank_s"]) == ranks
    assert c4["identical_to_sequential_block"] and c4["pdus"] >= 15 and c4["fallbacks"] == 0
    assert c4["packet_straddles_first_cut"]
    # no rank but the one that chains (rank 0 may walk a segment again) touched the whole capture: its own segment and the overlap
    assert max(c4["samples_spanned_per_rank"][1:]) < 0.9 * c4["samples"]
    print(line)
