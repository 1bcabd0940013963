"""CPU tests of the boundary: the C-ABI library loads, exports every symbol the headers declare,
its host-side table builders reproduce the golden known answers, and argument checking works.
No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    import gr_lora_amd
    return gr_lora_amd


def declared_symbols():
    names = []
    for h in ("lora_hip_demod.h", "lora_demod_core.h", "lora_hip_chan.h", "lora_decode.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"LORA_(?:HIP|DECODE)_API\s+[^;(]*?\b(lora_\w+)\s*\(", txt)
    return names


def test_library_exports_every_declared_symbol(g):
    lib = g.capi.lib()
    decl = declared_symbols()
    assert len(decl) >= 22
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert set(decl) == set(g.capi.EXPORTS)


def test_no_oracle_in_product(g):
    """The product must never route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gr-lora_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "lora_oracle" not in txt \
                    and "chan_oracle" not in txt, fn
    out = os.popen(f"ldd {g.capi.LIB_PATH}").read()
    assert "oracle" not in out


@pytest.mark.parametrize("sf", range(7, 13))
def test_host_table_builders_match_golden(sf, g, golden_index):
    kat = golden_index["tables"][str(sf)]
    up, down = g.make_chirp_tables(sf)
    w = g.make_kaiser(1 << sf, cases.BETA)
    assert cases.sha256(up) == kat["up_sha256"]
    assert cases.sha256(down) == kat["down_sha256"]
    assert cases.sha256(w) == kat["window_sha256"]


def test_argument_checking_without_gpu(g):
    lib = g.capi.lib()
    assert lib.lora_hip_strerror(0) == b"ok"
    assert b"fft_factor" in lib.lora_hip_strerror(g.capi.EUNSUPPORTED)
    h = C.c_void_p()
    assert lib.lora_hip_demod_create(C.byref(h), 0, 5, 2, 25.0, None, None, None) == g.capi.EINVAL
    assert lib.lora_hip_demod_create(C.byref(h), 0, 13, 2, 25.0, None, None, None) == g.capi.EINVAL
    assert lib.lora_hip_demod_create(C.byref(h), 0, 8, 0, 25.0, None, None, None) == g.capi.EINVAL
    # any fft_factor is taken as long as 2^sf * fft_factor fits the reference's unsigned short d_fft_size
    assert lib.lora_hip_demod_create(C.byref(h), 0, 12, 16, 25.0, None, None, None) == g.capi.EUNSUPPORTED
    assert lib.lora_hip_demod_create(C.byref(h), 0, 8, 256, 25.0, None, None, None) == g.capi.EUNSUPPORTED
    assert lib.lora_hip_demod_create(None, 0, 8, 2, 25.0, None, None, None) == g.capi.EINVAL
    up = np.zeros(4, np.complex64)
    assert lib.lora_hip_make_chirp_tables(3, up.ctypes.data, up.ctypes.data) == g.capi.EINVAL
    assert lib.lora_hip_make_kaiser(1, 25.0, up.ctypes.data) == g.capi.EINVAL
    # NULL handles are rejected, not dereferenced
    assert lib.lora_hip_demod_batch(None, None, 0, None, 0, None, None, 0) == g.capi.EINVAL
    assert lib.lora_hip_demod_batch_strided(None, None, 0, 0, 1, 0, 0, 0, 0, None, None, 0) == g.capi.EINVAL
    assert lib.lora_hip_demod_stage_iq(None, None, 0) == g.capi.EINVAL
    assert lib.lora_demod_core_work(None, None, 0, g.capi.PDU_CB(lambda *a: None), None) == g.capi.EINVAL
    lib.lora_hip_demod_destroy(None); lib.lora_demod_core_destroy(None)


def test_python_mirror_argument_checks(g):
    with pytest.raises(ValueError):
        g.demod(5, False, 25.0, 2)
    with pytest.raises(ValueError):
        g.demod(8, False, 25.0, 0)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(g.LoraHipError) as e:
            g.demod(8, False, 25.0, 2)          # no GPU: fails loudly, never falls back to a CPU path
        assert e.value.code == g.capi.ENODEVICE


def test_synth_matches_mod_layout(g):
    """gr-lora_amd/synth.py restates the mod frame (lib/mod_impl.cc:80-139); same bits as the oracle's."""
    import importlib
    import oracle
    synth = importlib.import_module("gr-lora_amd.synth")
    for sf in (7, 9):
        syms = oracle.random_symbols(sf, 11, 5)
        assert np.array_equal(synth.modulate(sf, syms, 0x12), oracle.modulate(sf, syms, 0x12))


def test_channeliser_piece_limit_keeps_positions_in_32_bits(g):
    """Fallback schedule of the channeliser (rate > nfilts): a position is i_in*nfilts + j with j < nfilts and is kept
    in 32 bits, so the pieces work() cuts a long input into must satisfy nfilts * piece <= 2^31 for every nfilts that
    lora_hip_chan_create accepts (1..1024)."""
    lib = g.capi.lib()
    for nfilts in list(range(1, 70)) + [127, 128, 500, 1023, 1024]:
        piece = lib.lora_hip_chan_piece_limit(nfilts)
        assert piece >= 1 << 21
        assert nfilts * piece <= 1 << 31
        assert nfilts * (piece + 1) > (1 << 31) - nfilts       # and no smaller than it has to be
