"""BASELINE.json configs[3] and configs[4] at FULL size on the GPU (tools/run_configs.py), so that the round's GPU test
record carries them and not only a builder's log:
  configs[3]: 1024 independent SF10 captures (2.2 GiB of IQ from host memory), 8 simulated ranks, four blocks in flight per
              rank; every 16th capture is checked against the oracle.
  configs[4]: the 1-hour SF9 capture (4.5e8 samples, 3.35 GiB, Poisson arrivals), sequential block, design-A time sharding
              into 8 segments, and the oracle on one host core: all three must give the same PDUs.
The oracle is the checker only."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def run_configs():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import run_configs
    return run_configs


@pytest.mark.timeout(400)
def test_configs3_full_size(run_configs):
    r = run_configs.cfg3(threads=4)
    assert r["captures"] == 1024 and r["pdus"] == 1024
    assert r["checked"] == 64 and r["differ"] == 0


@pytest.mark.timeout(600)
def test_configs4_full_size(run_configs):
    r = run_configs.cfg4()
    assert r["packets"] > 2000 and abs(r["pdus"] - r["packets"]) <= 10       # packets cut by the capture's end aside
    assert r["design_a_identical"] and r["sharded_identical"] and r["fallbacks"] == 0
    assert r["oracle_identical"], f"{r['oracle_pdus_differ']} PDUs differ from the oracle"
