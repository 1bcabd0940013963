"""ctypes binding of the CPU oracle (oracle/lora_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gr-lora_amd/) never does.  PARITY UNPINNED: see oracle/lora_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblora_oracle.so")

# include/lora/demod.h:38-46
S_RESET, S_PREFILL, S_DETECT_PREAMBLE, S_SFD_SYNC, S_READ_HEADER, S_READ_PAYLOAD, S_OUT = range(7)
KIND_UP, KIND_SFD = 0, 1

TRACE_DTYPE = np.dtype([("pos", "<u8"), ("state", "<u2"), ("chirp_off", "<u2"), ("idx", "<u2"),
                        ("consumed", "<u2"), ("val", "<f4"), ("n_sfd", "<u4")], align=True)


def build(force=False):
    """Compile oracle/liblora_oracle.so with the committed Makefile (gcc only)."""
    srcs = [os.path.join(_HERE, n) for n in ("lora_oracle.c", "lora_oracle.h", "chan_oracle.c", "chan_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(x) for x in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblora_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    vp, u, i, f, d, sz = C.c_void_p, C.c_uint, C.c_int, C.c_float, C.c_double, C.c_size_t
    L.lo_chirp_tables.argtypes = [u, vp, vp]
    L.lo_kaiser.argtypes = [u, d, vp]
    L.lo_mod_frame_len.argtypes = [u, sz]; L.lo_mod_frame_len.restype = sz
    L.lo_modulate.argtypes = [u, u, vp, sz, vp]
    L.lo_awgn.argtypes = [vp, sz, d, C.c_uint64, f]
    L.lo_splitmix64.argtypes = [C.POINTER(C.c_uint64)]; L.lo_splitmix64.restype = C.c_uint64
    L.lo_demod_create.argtypes = [u, i, f, u]; L.lo_demod_create.restype = vp
    L.lo_demod_destroy.argtypes = [vp]
    for n in ("lo_demod_upchirp", "lo_demod_downchirp", "lo_demod_window"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = vp
    L.lo_window_argmax.argtypes = [vp, vp, u, i, C.POINTER(f), C.POINTER(d)]
    L.lo_window_argmax.restype = u
    L.lo_window_argmax_batch.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, vp, vp]
    L.lo_demod_stream.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.lo_demod_stream.restype = sz
    L.lo_demod_time.argtypes = [u, i, f, u, vp, sz, u, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.lo_demod_time.restype = d
    L.lo_demod_time_opt.argtypes = [u, i, f, u, vp, sz, u, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    L.lo_demod_time_opt.restype = d
    # chan_oracle.h
    L.lo_chan_design_taps.argtypes = [d, u, d, vp, sz]; L.lo_chan_design_taps.restype = sz
    L.lo_rotator.argtypes = [d, vp, vp, sz, vp, C.POINTER(u)]
    L.lo_pfb_create.argtypes = [f, vp, sz, u]; L.lo_pfb_create.restype = vp
    L.lo_pfb_destroy.argtypes = [vp]
    for n in ("lo_pfb_taps_per_filter", "lo_pfb_dec_rate", "lo_pfb_last_filter"):
        getattr(L, n).argtypes = [vp]; getattr(L, n).restype = u
    L.lo_pfb_flt_rate.argtypes = [vp]; L.lo_pfb_flt_rate.restype = f
    L.lo_pfb_work.argtypes = [vp, vp, sz, vp, sz, i]; L.lo_pfb_work.restype = sz
    L.lo_pfb_schedule.argtypes = [vp, sz, vp, vp, sz]; L.lo_pfb_schedule.restype = sz
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def chirp_tables(sf):
    """(up, down), 2N complex64 each -- demod_impl.cc:92-101."""
    n = 1 << sf
    up = np.empty(2 * n, np.complex64); down = np.empty(2 * n, np.complex64)
    lib().lo_chirp_tables(sf, _ptr(up), _ptr(down))
    return up, down


def kaiser(ntaps, beta):
    w = np.empty(ntaps, np.float32)
    lib().lo_kaiser(ntaps, float(beta), _ptr(w))
    return w


def modulate(sf, symbols, sync_word=0x12):
    """One mod frame: 4N zeros, preamble, sync, SFD, payload, 4N+128 zeros (mod_impl.cc:80-139)."""
    syms = np.ascontiguousarray(symbols, np.uint16)
    out = np.empty(lib().lo_mod_frame_len(sf, len(syms)), np.complex64)
    lib().lo_modulate(sf, sync_word, _ptr(syms), len(syms), _ptr(out))
    return out


def awgn(iq, sigma2, seed, gain=1.0):
    out = np.array(iq, np.complex64, copy=True)
    lib().lo_awgn(_ptr(out), len(out), float(sigma2), int(seed), float(gain))
    return out


def splitmix64_stream(seed, count):
    st = C.c_uint64(seed)
    return np.array([lib().lo_splitmix64(C.byref(st)) for _ in range(count)], np.uint64)


def random_symbols(sf, n_sym, seed):
    """Payload symbols uniform in [0,N); the first 8 are multiples of 4."""
    r = splitmix64_stream(seed, n_sym)
    n = 1 << sf
    s = (r % np.uint64(n)).astype(np.uint16)
    s[:8] &= np.uint16(0xFFFC)
    return s


class Demod:
    """Oracle instance mirroring gr::lora::demod::make(sf, low_data_rate, beta, fft_factor)."""

    def __init__(self, spreading_factor=8, low_data_rate=False, beta=25.0, fft_factor=2):
        self.sf, self.ldr, self.beta, self.fft_factor = spreading_factor, bool(low_data_rate), beta, fft_factor
        self.N = 1 << spreading_factor
        self.M = self.N * fft_factor
        self._h = lib().lo_demod_create(spreading_factor, int(low_data_rate), beta, fft_factor)
        if not self._h:
            raise ValueError("bad demod parameters")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_demod_destroy(self._h)
            self._h = None

    def _view(self, fn, count, dtype):
        p = getattr(lib(), fn)(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_byte)), (count * np.dtype(dtype).itemsize,)) \
            .view(dtype).copy()

    @property
    def upchirp(self):
        return self._view("lo_demod_upchirp", 2 * self.N, np.complex64)

    @property
    def downchirp(self):
        return self._view("lo_demod_downchirp", 2 * self.N, np.complex64)

    @property
    def window(self):
        return self._view("lo_demod_window", self.N, np.float32)

    def window_argmax(self, iq, pos, chirp_off, kind, want_margin=True):
        """L1 batch.  Returns (idx u16, val f32, margin f64 or None)."""
        iq = np.ascontiguousarray(iq, np.complex64)
        pos = np.ascontiguousarray(pos, np.uint64)
        n = len(pos)
        off = np.ascontiguousarray(np.broadcast_to(chirp_off, (n,)), np.uint16)
        kd = np.ascontiguousarray(np.broadcast_to(kind, (n,)), np.uint8)
        idx = np.empty(n, np.uint16); val = np.empty(n, np.float32)
        mar = np.empty(n, np.float64) if want_margin else None
        lib().lo_window_argmax_batch(self._h, _ptr(iq), len(iq), _ptr(pos), _ptr(off), _ptr(kd), n,
                                     _ptr(idx), _ptr(val), _ptr(mar) if want_margin else None)
        return idx, val, mar

    def stream(self, capture, want_trace=False):
        """L2: run the block over one finite capture.  Returns (pdus, trace, n_calls, n_ffts)."""
        cap = np.ascontiguousarray(capture, np.complex64)
        max_calls = len(cap) // max(1, self.N // 16) + 16
        max_syms = len(cap) // self.N + 16
        syms = np.empty(max_syms, np.uint16)
        plen = np.empty(max_syms, np.uint32)
        trace = np.zeros(max_calls if want_trace else 1, TRACE_DTYPE)
        nc, nf = C.c_size_t(0), C.c_size_t(0)
        npdu = lib().lo_demod_stream(self._h, _ptr(cap), len(cap), _ptr(syms), max_syms, _ptr(plen), max_syms,
                                     _ptr(trace) if want_trace else None, max_calls if want_trace else 0,
                                     C.byref(nc), C.byref(nf))
        pdus, k = [], 0
        for i in range(npdu):
            pdus.append(syms[k:k + plen[i]].copy()); k += int(plen[i])
        return pdus, (trace[:nc.value] if want_trace else None), nc.value, nf.value


def _stream_framed(self, framed, pos0=0):
    """The block over an already framed stream whose first sample is framed-stream index pos0, entered in S_RESET (what
    the product's lora_demod_core_seek does).  Returns (pdus, trace) with trace positions in framed-stream indices."""
    x = np.ascontiguousarray(framed, np.complex64)
    max_calls = len(x) // max(1, self.N // 16) + 16
    max_syms = len(x) // self.N + 16
    syms = np.empty(max_syms, np.uint16); plen = np.empty(max_syms, np.uint32)
    trace = np.zeros(max_calls, TRACE_DTYPE)
    nc = C.c_size_t(0)
    fn = lib().lo_demod_stream_framed
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                   C.POINTER(C.c_size_t)]
    npdu = fn(self._h, _ptr(x), len(x), _ptr(syms), max_syms, _ptr(plen), max_syms, _ptr(trace), max_calls, C.byref(nc))
    pdus, k = [], 0
    for i in range(npdu):
        pdus.append(syms[k:k + plen[i]].copy()); k += int(plen[i])
    tr = trace[:nc.value].copy()
    tr["pos"] += np.uint64(pos0)
    return pdus, tr


Demod.stream_framed = _stream_framed


def time_stream(sf, ldr, beta, fft_factor, capture, reps=1, optimised=False):
    """cpu_baseline leg: seconds for `reps` passes of the faithful loop (or of its optimised variant);
    returns (sec, calls, ffts, pdus) per pass."""
    cap = np.ascontiguousarray(capture, np.complex64)
    nc, nf, npd = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    fn = lib().lo_demod_time_opt if optimised else lib().lo_demod_time
    sec = fn(sf, int(ldr), beta, fft_factor, _ptr(cap), len(cap), reps,
             C.byref(nc), C.byref(nf), C.byref(npd))
    return sec, nc.value, nf.value, npd.value


# ---- channeliser (chan_oracle.h): rotator_cc + pfb_arb_resampler_ccf, examples/rx_usrp.grc:489-503, :825-860 ----
def design_taps(rate, nfilts=32, atten=100.0):
    """pfb.arb_resampler_ccf.create_taps for rate < 1 (firdes.low_pass_2, Blackman-harris)."""
    n = lib().lo_chan_design_taps(float(rate), nfilts, float(atten), None, 0)
    t = np.empty(n, np.float32)
    lib().lo_chan_design_taps(float(rate), nfilts, float(atten), _ptr(t), n)
    return t


class Rotator:
    """blocks.rotator_cc(phase_inc): float phase recurrence, renormalised every 512 samples."""

    def __init__(self, phase_inc):
        self.phase_inc = float(phase_inc)
        self._phase = np.array([1 + 0j], np.complex64)
        self._counter = C.c_uint(0)

    def work(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.empty_like(x)
        lib().lo_rotator(self.phase_inc, _ptr(x), _ptr(out), len(x), _ptr(self._phase), C.byref(self._counter))
        return out


class PfbArbResampler:
    """filter.pfb_arb_resampler_ccf(rate, taps, nfilts) on flat arrays, state carried between calls."""

    def __init__(self, rate, taps=None, nfilts=32, atten=100.0):
        self.rate, self.nfilts = float(rate), nfilts
        self.taps = design_taps(rate, nfilts, atten) if taps is None else np.ascontiguousarray(taps, np.float32)
        self._h = lib().lo_pfb_create(self.rate, _ptr(self.taps), len(self.taps), nfilts)
        if not self._h:
            raise ValueError("bad resampler parameters")
        self.taps_per_filter = lib().lo_pfb_taps_per_filter(self._h)
        self.dec_rate = lib().lo_pfb_dec_rate(self._h)
        self.flt_rate = lib().lo_pfb_flt_rate(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lo_pfb_destroy(self._h)
            self._h = None

    def max_output(self, n_in):
        return int(int(n_in) * self.rate * (1 + 1e-6)) + 64        # one output per (dec_rate + flt_rate) positions

    def work(self, x, use_double=False):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.empty(self.max_output(len(x)), np.complex64)
        n = lib().lo_pfb_work(self._h, _ptr(x), len(x), _ptr(out), len(out), int(use_double))
        return out[:n].copy()

    def schedule(self, n_in):
        """(32*i_in + j, d_acc) per output for the next n_in inputs; advances the state like work()."""
        cap = self.max_output(n_in)
        pos = np.empty(cap, np.uint64); acc = np.empty(cap, np.float32)
        n = lib().lo_pfb_schedule(self._h, n_in, _ptr(pos), _ptr(acc), cap)
        return pos[:n].copy(), acc[:n].copy()
