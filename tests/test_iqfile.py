"""Raw float32 I/Q capture files (gr-lora_amd/iqfile.py): layout and chunked streaming."""
import importlib

import numpy as np
import pytest

import cases
import oracle

iqfile = importlib.import_module("gr-lora_amd.iqfile")


def test_layout_is_interleaved_float32(tmp_path):
    x = (np.arange(10) + 1j * np.arange(10, 20)).astype(np.complex64)
    p = tmp_path / "c.32iq"
    iqfile.write_iq(p, x)
    raw = np.fromfile(p, dtype="<f4")
    assert raw.tolist() == [v for k in range(10) for v in (float(k), float(10 + k))]
    assert np.array_equal(iqfile.read_iq(p), x) and np.array_equal(iqfile.read_iq(p, mmap=False), x)
    assert sum(len(c) for c in iqfile.iter_chunks(p, 3)) == 10


class _OracleBlock:
    spreading_factor = 7

    def __init__(self):
        self.d = oracle.Demod(7, False, cases.BETA, 2)
        self.buf = np.zeros(0, np.complex64)

    def work(self, samples):           # the oracle has no streaming entry: accumulate, demodulate at the end
        self.buf = np.concatenate([self.buf, samples])
        return []


def test_chunked_feed_appends_the_flush_zeros(tmp_path):
    cap, _, _ = cases.make_capture(7, [10], seed=3, tail_pad_symbols=0)
    p = tmp_path / "c.32iq"
    iqfile.write_iq(p, cap)
    blk = _OracleBlock()
    iqfile.demod_file(p, blk, chunk_samples=1000)
    assert len(blk.buf) == len(cap) + 4 * 128 and np.array_equal(blk.buf[:len(cap)], cap) and not blk.buf[len(cap):].any()
    assert len(blk.d.stream(blk.buf)[0]) == 1          # the packet is flushed thanks to the appended zeros
    assert len(blk.d.stream(cap)[0]) == 0              # and would not be without them (GNU Radio framing)


@pytest.mark.gpu
def test_gpu_file_equals_array(tmp_path):
    import gr_lora_amd as g
    cap, _, _ = cases.make_capture(8, [20, 31, 12], seed=8, tail_pad_symbols=0)
    p = tmp_path / "c.32iq"
    iqfile.write_iq(p, cap)
    blk = g.demod(8, False, cases.BETA, 2)
    got = iqfile.demod_file(p, blk, chunk_samples=7777)
    blk.close()
    ref = oracle.Demod(8, False, cases.BETA, 2).stream(np.concatenate([cap, np.zeros(4 * 256, np.complex64)]))[0]
    assert [x.tolist() for x in got] == [x.tolist() for x in ref] and len(ref) == 3
