"""Randomised differential test of the whole block against the oracle (generator: tools/fuzz_l2.py).
400 seeded cases here; runs of up to 15 minutes (52 892 cases, 5.07 M calls, 86 676 PDUs) are recorded in
profiles/r01_fuzz.txt."""
import os
import sys

import numpy as np
import pytest

from test_gpu_l2_parity import compare_with_oracle

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def test_random_captures_follow_the_oracle():
    import gr_lora_amd as g
    from fuzz_l2 import random_case
    rng = np.random.default_rng(20240)
    same = div = pdus = 0
    for k in range(400):
        sf, f, ldr, cap, chunks = random_case(rng)
        r = compare_with_oracle(g, sf, f, ldr, cap, chunks=chunks, name=f"fuzz {k} sf{sf} f{f} ldr{ldr}")
        same += r["same_path"]; div += not r["same_path"]; pdus += r["pdus"]
    print(f"{same} identical call paths, {div} sub-margin divergences, {pdus} PDUs")
    assert same >= 380 and pdus >= 300

