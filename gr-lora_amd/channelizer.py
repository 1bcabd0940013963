"""Host-side mirror of the two stock GNU Radio blocks the reference puts in front of lora.demod
(examples/rx_usrp.grc:489-503 blocks_rotator_cc, :825-860 pfb_arb_resampler_ccf), over the C ABI of
include/lora_hip_chan.h.  Product code: no CPU fallback, nothing here touches oracle/.

    ch = channelizer(phase_inc=2*math.pi*offset/samp_rate, rrate=bw/samp_rate)    # taps="" in the flowgraph
    y = ch.work(x)            # numpy complex64 in -> numpy complex64 out; torch cuda tensors stay on the device
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import LoraHipError, lib


def create_taps(rate, flt_size=32, atten=100.0):
    """pfb.arb_resampler_ccf.create_taps (gr-filter python/pfb.py), rate < 1."""
    n = C.c_size_t(0)
    rc = lib().lora_hip_chan_design_taps(float(rate), flt_size, float(atten), None, 0, C.byref(n))
    if rc:
        raise LoraHipError(rc)
    t = np.empty(n.value, np.float32)
    rc = lib().lora_hip_chan_design_taps(float(rate), flt_size, float(atten), t.ctypes.data, len(t), C.byref(n))
    if rc:
        raise LoraHipError(rc)
    return t


class channelizer:
    """blocks.rotator_cc(phase_inc) -> pfb.arb_resampler_ccf(rrate, taps, nfilts, atten) as one GPU block.
    Argument names follow the GRC block parameters (rx_usrp.grc:501, :837-854)."""

    def __init__(self, phase_inc=0.0, rrate=1.0, taps=None, nfilts=32, atten=100.0, device=0):
        self.phase_inc, self.rrate, self.nfilts, self.device = float(phase_inc), float(rrate), int(nfilts), int(device)
        self.taps = create_taps(rrate, nfilts, atten) if taps is None or len(taps) == 0 \
            else np.ascontiguousarray(taps, np.float32)
        self._h = C.c_void_p()
        self._destroy = lib().lora_hip_chan_destroy
        rc = lib().lora_hip_chan_create(C.byref(self._h), device, self.phase_inc, self.rrate, self.taps.ctypes.data,
                                        len(self.taps), self.nfilts)
        if rc:
            self._h = None
            raise LoraHipError(rc)

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise LoraHipError(rc, lib().lora_hip_chan_last_hip_error(self._h).decode())

    taps_per_filter = property(lambda self: lib().lora_hip_chan_taps_per_filter(self._h))
    dec_rate = property(lambda self: lib().lora_hip_chan_dec_rate(self._h))
    flt_rate = property(lambda self: lib().lora_hip_chan_flt_rate(self._h))
    kernel_name = property(lambda self: lib().lora_hip_chan_kernel_name(self._h).decode())
    phase_inc_effective = property(lambda self: lib().lora_hip_chan_phase_inc_effective(self._h))

    def force_general(self, on=True):
        self._check(lib().lora_hip_chan_force_general(self._h, int(on)))

    def set_stream(self, stream_ptr):
        """Run on a caller-owned hipStream_t; None/0 goes back to the default: the handle's own stream for host
        arrays, torch's current stream for cuda tensors (so that work() is ordered after whatever produced them)."""
        self._user_stream = bool(stream_ptr)
        self._check(lib().lora_hip_chan_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def reset(self):
        self._check(lib().lora_hip_chan_reset(self._h))

    def seek(self, n_consumed, history=None):
        """lora_hip_chan_seek: continue as if n_consumed samples of the stream had been processed; `history` = the
        taps_per_filter - 1 samples before that position (numpy or torch cuda tensor; None = zeros)."""
        ptr, flags = None, 0
        if history is not None:
            k1 = self.taps_per_filter - 1
            if capi._is_torch(history):
                assert history.is_cuda and history.is_contiguous() and history.numel() == k1
                import torch
                torch.cuda.current_stream(history.device).synchronize()
                ptr, flags = history.data_ptr(), capi.IQ_ON_DEVICE
            else:
                history = np.ascontiguousarray(history, np.complex64)
                assert len(history) == k1
                ptr = history.ctypes.data
        self._check(lib().lora_hip_chan_seek(self._h, int(n_consumed), ptr, flags))

    def output_count(self, n_in):
        return lib().lora_hip_chan_output_count(self._h, int(n_in))

    def work(self, x, out=None, async_=False):
        """One work() call over x.  numpy in -> numpy out; a torch cuda complex64 tensor in -> torch out
        (pass `out` to reuse a buffer).  Returns the outputs produced (a view of `out` when given)."""
        on_dev = capi._is_torch(x) and x.is_cuda
        n_in = x.numel() if capi._is_torch(x) else len(x)
        n_out = self.output_count(n_in)
        if on_dev:
            import torch
            if not getattr(self, "_user_stream", False):
                cur = torch.cuda.current_stream(x.device)
                if cur.cuda_stream:
                    self._check(lib().lora_hip_chan_set_stream(self._h, C.c_void_p(cur.cuda_stream)))
                else:
                    # torch's default stream is the null stream, which the C ABI cannot name (NULL = the handle's own
                    # stream): order by hand -- wait for the producer of x, run synchronously on the handle's stream
                    cur.synchronize()
                    self._check(lib().lora_hip_chan_set_stream(self._h, None))
                    async_ = False
            if out is None:
                out = torch.empty(max(n_out, 1), dtype=torch.complex64, device=x.device)
            assert out.is_cuda and out.is_contiguous() and x.is_contiguous() and out.numel() >= n_out
            pin, pout, cap = x.data_ptr(), out.data_ptr(), out.numel()
            flags = capi.IQ_ON_DEVICE | capi.OUT_ON_DEVICE | (capi.ASYNC if async_ else 0)
        else:
            x = np.ascontiguousarray(x, np.complex64)
            if not getattr(self, "_user_stream", False):
                self._check(lib().lora_hip_chan_set_stream(self._h, None))
            if out is None:
                out = np.empty(max(n_out, 1), np.complex64)
            pin, pout, cap, flags = x.ctypes.data, out.ctypes.data, len(out), 0
        got = C.c_size_t(0)
        self._check(lib().lora_hip_chan_work(self._h, pin, n_in, pout, cap, C.byref(got), flags))
        return out[:got.value]
