"""BASELINE.json `configs` other than the bench workload, in small, as parity cases on the GPU.

configs[0]  txrx_sim loopback: SF8, fft_factor 2, beta 25, 64-byte PDUs (= 128 symbols after the
            reference's encoder at CR 4/8 with explicit header; the symbols are fed directly, the
            encoder is outside the path), low_data_rate False (BASELINE text) and True (shipped grc).
configs[3]  many independent channelised captures, SF10, sharded over ranks (here: 12 captures, 3 simulated
            ranks in one process; the 2-rank gloo version of the same code runs on the CPU).
configs[4]  one long capture with random packet arrivals, SF9, time-sharded (design A); here 40 s with
            exponential gaps, 4 segments, against the sequential block and against the oracle.
The comparison with the oracle uses the L2 bar (tests/test_gpu_l2_parity.py)."""
import importlib

import numpy as np
import pytest

import cases
import oracle
from test_gpu_l2_parity import compare_with_oracle

pytestmark = pytest.mark.gpu
shard = importlib.import_module("gr-lora_amd.shard")


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


@pytest.mark.parametrize("ldr", [0, 1])
def test_cfg0_txrx_loopback_128_symbol_pdus(g, ldr):
    cap, tx, _ = cases.make_capture(8, [128, 128, 128], seed=1)
    r = compare_with_oracle(g, 8, 2, ldr, cap, name=f"cfg0 ldr{ldr}")
    assert r["pdus"] == 3
    # the reference's own loopback property: header symbols come back /4, payload within its +-2 bin jitter
    blk = g.demod(8, bool(ldr), cases.BETA, 2)
    pdus = blk.work(cap)
    blk.close()
    for p, t in zip(pdus, tx):
        assert len(p) == len(t) + 1                                  # trailing squelch symbol (a14)
        d = (p[8:len(t)].astype(int) - (t[8:].astype(int) // (4 if ldr else 1))) % (256 // (4 if ldr else 1))
        d = np.minimum(d, (256 // (4 if ldr else 1)) - d)
        assert d.max() <= 2


def test_cfg3_many_captures_sharded(g):
    sf, f, n_cap, world = 10, 2, 12, 3

    def load(cid):
        return cases.make_capture(sf, [24 + cid % 5], seed=4000 + cid)[0]

    merged = {}
    for rank in range(world):
        lo, hi = shard.my_range(n_cap, rank, world)
        local = shard.demod_captures(range(lo, hi), load, lambda: g.demod(sf, False, cases.BETA, f))
        assert set(local) == set(range(lo, hi))
        merged.update(local)
    out = shard.gather_pdus(merged, n_cap)                           # no process group: single-rank merge
    d = oracle.Demod(sf, False, cases.BETA, f)
    n_pdus = 0
    for cid in range(n_cap):
        ref = d.stream(load(cid))[0]
        assert [p.tolist() for p in out[cid]] == [p.tolist() for p in ref], cid
        n_pdus += len(ref)
    assert n_pdus >= n_cap


def _poisson_capture(sf, seconds, mean_gap_s, seed):
    """Packets of 16..128 symbols separated by exponential gaps (arbitrary sample counts, so every lattice
    phase occurs), 125 kS/s."""
    rng = np.random.default_rng(seed)
    n, total = 1 << sf, int(125e3 * seconds)
    parts, used, tx = [], 0, 0
    while True:
        gap = int(rng.exponential(mean_gap_s) * 125e3)
        ns = int(rng.integers(16, 129))
        frame = oracle.modulate(sf, oracle.random_symbols(sf, ns, seed * 7919 + tx), cases.SYNC_WORD)
        if used + gap + len(frame) + 4 * n > total:
            break
        parts += [np.zeros(gap, np.complex64), frame]
        used += gap + len(frame); tx += 1
    parts.append(np.zeros(total - used, np.complex64))
    return np.concatenate(parts), tx


def test_cfg4_long_capture_time_sharded(g):
    sf, f, world = 9, 2, 4
    cap, n_tx = _poisson_capture(sf, 40.0, 1.0, seed=5)
    assert n_tx >= 15
    # sequential block against the oracle (L2 bar)
    r = compare_with_oracle(g, sf, f, 0, cap, name="cfg4 sequential")
    assert r["pdus"] >= n_tx - 2
    seq = g.demod(sf, False, cases.BETA, f)
    seq.enable_trace(len(cap) // (1 << sf) * 2 + 64)
    ref_pdus = seq.work(cap)
    ref_tr = seq.trace().copy()
    seq.close()
    # design A: 4 time segments swept independently, merged, replayed
    sweep = shard.HipSweeper(sf, f, cases.BETA)
    parts = [shard.sweep_segment(sweep, cap, sf, q0, q1) for q0, q1 in shard.lattice_partition(len(cap), sf, world)]
    sweep.close()
    blk = g.demod(sf, False, cases.BETA, f)
    blk.enable_trace(len(ref_tr) + 64)
    for start, idx, val in shard.merge_sweeps(parts, sf):
        blk.preload(start, idx, val)
    pdus = blk.work(cap)
    tr = blk.trace()
    blk.close()
    assert [p.tolist() for p in pdus] == [p.tolist() for p in ref_pdus]
    assert len(tr) == len(ref_tr)
    for fld in ("pos", "state", "chirp_off", "idx", "consumed", "n_sfd"):
        assert np.array_equal(tr[fld], ref_tr[fld]), fld
    assert np.array_equal(tr["val"].view(np.uint32), ref_tr["val"].view(np.uint32))
    # every lattice phase was exercised by the random gaps
    locks = ref_tr["pos"][ref_tr["state"] == 4] if (ref_tr["state"] == 4).any() else ref_tr["pos"][:0]
    assert len(set((locks % (1 << sf)).tolist())) >= 4


@pytest.mark.parametrize("ldr", [True, False])
def test_cfg0_bytes_in_bytes_out(g, ldr):
    """BASELINE configs[0] as the flowgraph runs it (examples/txrx_sim.grc:686-733): 64-byte PDUs -> encode -> mod ->
    demod -> decode -> bytes.  encode and mod are the oracle's restatements (test infrastructure); demod is the GPU block,
    decode the C ABI of include/lora_decode.h.  The chain on the GPU must give exactly the bytes the oracle's chain
    (oracle demod -> restated decode) gives."""
    from oracle import codec_oracle as codec
    rng = np.random.default_rng(64 + ldr)
    w = rng.integers(0, 64 if ldr else 256, 400).astype(np.uint16)
    w[:8] &= 0x3F
    payloads = [rng.integers(0, 256, 64).astype(np.uint8) for _ in range(3)]
    frames, tx = [], []
    for data in payloads:
        s = np.array(codec.encode_pdu(8, 4, ldr, w.tolist(), data.tolist()), np.uint16)
        if not ldr:
            s[8:] &= 0xFF
        tx.append(s)
        frames.append(oracle.modulate(8, s, cases.SYNC_WORD))
    frames.append(np.zeros(4 * 256, np.complex64))
    cap = np.concatenate(frames)
    blk = g.demod(8, ldr, cases.BETA, 2)
    pdus = blk.work(cap)
    blk.close()
    ref_pdus = oracle.Demod(8, ldr, cases.BETA, 2).stream(cap)[0]
    assert len(pdus) == len(ref_pdus) == 3
    dec = g.decode(8, 4, ldr, False, whitening_sequence=w)
    exact = 0
    for p, r, data in zip(pdus, ref_pdus, payloads):
        assert np.array_equal(p, r)                                           # same symbols as the oracle's block
        out = bytes(dec.work(p))
        assert out == codec.decode_pdu(8, 4, ldr, w.tolist(), [int(v) for v in r])
        # 64 bytes are 128 nybbles: 6 go into the header block, the encoder's integer division keeps whole payload
        # blocks only (encode_impl.cc:200) and drops the last 2 nybbles: 63 bytes come back
        assert len(out) == 63
        exact += out == bytes(data)[:63]
    # Not asserted: exact > 0.  The reference's demod returns some payload symbols one bin low in a clean loopback
    # ("indicative behaviour"), which its Hamming code corrects only in part; the bar is the reference's
    # output (the oracle chain above), not the transmitted bytes.  Without the modem the chain is exact at ldr True
    # (tests/test_decode_chain.py).
    print(f"cfg0 ldr={ldr}: {exact}/3 PDUs equal the transmitted bytes")
