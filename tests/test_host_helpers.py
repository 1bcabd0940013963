"""CPU tests of two host-side pieces of the block that no GPU run pins down by itself.

  * The preamble rule of the planned block's SFD stage (demod_core.cpp: rule_hits_sse2 / rule_hits_avx2, reference
    lib/demod_impl.cc:247-273) through the test hook lora_demod_core_rule_hits: a position the vector form missed would
    not change a result -- the window would be evaluated on demand -- only cost a launch, so parity tests cannot see it.
  * PduList, the lazy sequence demod.work() returns."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


def rule_hits(lib, idx, val, thr, path):
    idx = np.ascontiguousarray(idx, np.uint16); val = np.ascontiguousarray(val, np.float32)
    hits = np.zeros(len(idx) + 1, np.uint32)
    n = lib.lora_demod_core_rule_hits(idx.ctypes.data, val.ctypes.data, C.c_float(thr), len(idx), hits.ctypes.data, path)
    if n == C.c_size_t(-1).value:
        return None
    return hits[:n].tolist()


def rule_definition(idx, val, thr):
    """:247-273 on the lattice: newest four windows of one phase (16 positions apart) within +-1 of the newest, not squelched"""
    i = idx.astype(np.int64)
    out = []
    for j in range(48, len(i)):
        if all(abs(i[j] - i[j - 16 * k]) <= 1 for k in (1, 2, 3)) and val[j] > thr:
            out.append(j)
    return out


def sweeps():
    rng = np.random.default_rng(11)
    yield np.zeros(0, np.uint16), np.zeros(0, np.float32)
    for n in list(range(40, 90)) + [127, 128, 129, 1000, 4099]:
        idx = np.where(rng.random(n) < 0.25, rng.integers(0, 8192, n), 100 + rng.integers(0, 3, n)).astype(np.uint16)
        val = np.where(rng.random(n) < 0.3, 0.0, rng.random(n) * 0.02).astype(np.float32)
        val[rng.random(n) < 0.02] = np.nan                    # a NaN maximum counts as squelched (val > thr is false)
        yield idx, val
    n = 3000                                                  # an hour of exact zeros: the same bin everywhere, all squelched
    yield np.zeros(n, np.uint16), np.zeros(n, np.float32)
    yield np.full(n, 7, np.uint16), np.full(n, 1.0, np.float32)     # ... and all above the threshold: every position holds
    idx = np.where(rng.random(n) < 0.5, 0, 8191).astype(np.uint16)  # bins at both ends: 0 and 8191 are not neighbours
    yield idx, np.full(n, 1.0, np.float32)
    idx = np.where(rng.random(n) < 0.5, 0, 65534).astype(np.uint16)  # nor are they in 16-bit wrap-around arithmetic
    yield idx, np.full(n, 1.0, np.float32)
    yield np.tile(np.arange(16, dtype=np.uint16) * 500, n // 16), np.full(n // 16 * 16, 1.0, np.float32)   # a preamble on every phase


def test_preamble_rule_vector_forms_equal_the_definition(g):
    lib = g.capi.lib()
    thr = 0.005
    ran_avx2 = False
    for idx, val in sweeps():
        want = rule_definition(idx, val, thr)
        assert rule_hits(lib, idx, val, thr, 3) == want
        assert rule_hits(lib, idx, val, thr, 1) == want, len(idx)
        assert rule_hits(lib, idx, val, thr, 0) == want, len(idx)
        got = rule_hits(lib, idx, val, thr, 2)
        if got is not None:
            ran_avx2 = True
            assert got == want, len(idx)
    print("AVX2 + BMI2 form ran:", ran_avx2)


def test_preamble_rule_on_a_synthetic_preamble(g):
    """eight equal bins in a row on one phase among noise bins: the rule holds from the fourth on (and nowhere else)"""
    lib = g.capi.lib()
    n = 16 * 40
    idx = (np.arange(n) * 37 % 8192).astype(np.uint16)         # no two bins 16, 32 or 48 positions apart are neighbours
    val = np.full(n, 0.01, np.float32)
    phase = 5
    for s in range(10, 18):
        idx[16 * s + phase] = 1234 + (s & 1)                   # +-1 wobble stays inside the tolerance
    want = [16 * s + phase for s in range(13, 18)]
    assert rule_definition(idx, val, 0.005) == want
    assert rule_hits(lib, idx, val, 0.005, 0) == want
    val[16 * 15 + phase] = 0.001                               # one of them squelched (:268)
    want.remove(16 * 15 + phase)
    assert rule_hits(lib, idx, val, 0.005, 0) == want


def test_pdu_list_behaves_as_the_list_it_stands_for():
    from importlib import import_module
    demod = import_module("gr-lora_amd.demod")
    flat = np.arange(10, dtype=np.uint16)
    pdus = demod.PduList(flat, [0, 3, 3, 10])
    ref = [flat[0:3], flat[3:3], flat[3:10]]
    assert len(pdus) == 3 and bool(pdus)
    for a, b in zip(pdus, ref):
        assert a.dtype == np.uint16 and np.array_equal(a, b)
    assert np.array_equal(pdus[-1], ref[-1]) and np.array_equal(pdus[0], ref[0])
    assert [p.tolist() for p in pdus[1:]] == [[], list(range(3, 10))]
    with pytest.raises(IndexError):
        pdus[3]
    assert pdus == [r.tolist() for r in ref] and pdus == ref
    assert not (pdus == ref[:2])
    acc = []
    acc += pdus                                               # how chunked feeds collect them
    assert len(acc) == 3 and len(pdus + pdus) == 6 and len([flat] + pdus) == 4
    empty = demod._EMPTY
    assert len(empty) == 0 and empty == [] and not empty and list(empty) == []
