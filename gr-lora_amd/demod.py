"""Python mirror of the reference's block interface for the demod path.

    lora.demod(spreading_factor, low_data_rate, beta, fft_factor)        grc/lora_demod.xml:7
    gr::lora::demod::make(unsigned short, bool, float, unsigned short)   include/lora/demod.h:66-69

Same argument order, meaning and defaults (SF 8, False, 25.0, 2: grc/lora_demod.xml:12-30).  The
block has one complex64 input stream and one message port "out" that carries
cons(make_dict(), u16vector) PDUs (lib/demod_impl.cc:58-60,74-75,420-422); here a PDU is the pair
({}, numpy uint16 array).  All numeric work runs on the GPU through the C ABI
(include/lora_demod_core.h -> include/lora_hip_demod.h); there is no CPU fallback.
"""
import ctypes as C

from collections.abc import Sequence

import numpy as np

from . import capi


class PduList(Sequence):
    """The PDUs of one work() call, in order: u16 arrays cut out of one copy of the block's symbol buffer when they are
    asked for (a few hundred numpy views cost more than the block's own host work on a dense capture).  Behaves as the
    list it stands for: len, indexing, slicing, iteration, +, == against a list."""
    __slots__ = ("_flat", "_off")

    def __init__(self, flat, off):
        self._flat, self._off = flat, off

    def __len__(self):
        return len(self._off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("PDU index out of range")
        return self._flat[self._off[i]:self._off[i + 1]]

    def __iter__(self):
        flat, off = self._flat, self._off
        for k in range(len(off) - 1):
            yield flat[off[k]:off[k + 1]]

    def __add__(self, other):
        return list(self) + list(other)

    def __radd__(self, other):
        return list(other) + list(self)

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(np.array_equal(a, b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    __hash__ = None

    def __repr__(self):
        return f"PduList({len(self)} PDUs)"


_EMPTY = PduList(np.zeros(0, np.uint16), [0])


class demod:
    """gr::lora::demod without GNU Radio: feed samples with work(), collect PDUs."""

    name = "demod"                       # gr::block("demod", ...), lib/demod_impl.cc:58
    out_port = "out"                     # pmt::mp("out"), lib/demod_impl.cc:74

    def __init__(self, spreading_factor=8, low_data_rate=False, beta=25.0, fft_factor=2, device=0, plan=None):
        # asserts of lib/demod_impl.cc:70-72 (compiled out in the reference's Release build)
        if not (5 < spreading_factor < 13):
            raise ValueError("spreading_factor must be in 6..12")
        if fft_factor <= 0:
            raise ValueError("fft_factor must be > 0")
        self.spreading_factor, self.low_data_rate = int(spreading_factor), bool(low_data_rate)
        self.beta, self.fft_factor = float(beta), int(fft_factor)
        self._h = C.c_void_p()
        self._destroy = capi.lib().lora_demod_core_destroy    # bound now: module globals are gone at interpreter exit
        rc = capi.lib().lora_demod_core_create(C.byref(self._h), device, self.spreading_factor,
                                               int(self.low_data_rate), self.beta, self.fft_factor)
        if rc:
            self._h = None
            raise capi.LoraHipError(rc)
        if plan is not None:                      # lora_demod_core_set_plan: 0 on demand, 1 long inputs, 2 always
            rc = capi.lib().lora_demod_core_set_plan(self._h, int(plan))
            if rc:
                raise capi.LoraHipError(rc)
        # PDUs are collected inside the block and fetched once per work() call: a ctypes callback per PDU costs more
        # than the block spends on a packet
        rc = capi.lib().lora_demod_core_set_collect(self._h, 1)
        if rc:
            raise capi.LoraHipError(rc)
        self._handlers = []
        self._pending = _EMPTY
        self._trace = None
        self._cb = capi.PDU_CB(0)                # NULL: no per-PDU callback

    @staticmethod
    def make(spreading_factor=8, low_data_rate=False, beta=25.0, fft_factor=2, device=0, plan=None):
        return demod(spreading_factor, low_data_rate, beta, fft_factor, device, plan)

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    def history(self):
        return 3 * (1 << self.spreading_factor)       # set_history(DEMOD_HISTORY_DEPTH*N), :103

    def forecast(self, noutput_items):
        return noutput_items * (1 << self.spreading_factor)    # :142-147

    def message_subscribe(self, handler):
        """handler(pdu) is called for every PDU published on port "out"."""
        self._handlers.append(handler)

    def _fetch_pdus(self):
        """The PDUs of the last work() call, as (meta, u16vector) pairs like pmt::cons(make_dict(), init_u16vector), :420-422;
        subscribed handlers see each of them, in order."""
        syms, offs, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        rc = capi.lib().lora_demod_core_collected(self._h, C.byref(syms), C.byref(offs), C.byref(n))
        if rc:
            raise capi.LoraHipError(rc)
        self._pending = _EMPTY
        if n.value:
            o = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_uint64)), (n.value + 1,))
            total = int(o[-1])
            flat = np.ctypeslib.as_array(C.cast(syms, C.POINTER(C.c_uint16)), (total,)).copy() if total else np.zeros(0, np.uint16)
            self._pending = PduList(flat, o.tolist())
        if self._handlers:
            for vec in self._pending:
                pdu = ({}, vec)
                for h in self._handlers:
                    h(pdu)

    def enable_trace(self, max_rows):
        self._trace = np.zeros(max_rows, capi.CORE_TRACE_DTYPE)
        capi.lib().lora_demod_core_set_trace(self._h, self._trace.ctypes.data, max_rows)

    def trace(self):
        n = min(int(capi.lib().lora_demod_core_calls(self._h)), len(self._trace))
        return self._trace[:n]

    def reset(self):
        """Start a new, independent stream (lora_demod_core_reset): same as a freshly made block."""
        rc = capi.lib().lora_demod_core_reset(self._h)
        if rc:
            raise capi.LoraHipError(rc)
        self._pending = _EMPTY

    def seek(self, pos):
        """lora_demod_core_seek: a new stream that starts at framed-stream index pos in S_RESET; the next samples fed
        are framed-stream samples from pos on (no history zeros in front)."""
        rc = capi.lib().lora_demod_core_seek(self._h, int(pos))
        if rc:
            raise capi.LoraHipError(rc)
        self._pending = _EMPTY

    def idle_log(self):
        """lora_demod_core_idle_log as an (k, 3) uint64 array: first index, last index, PDUs emitted before."""
        p, n = C.c_void_p(), C.c_size_t(0)
        rc = capi.lib().lora_demod_core_idle_log(self._h, C.byref(p), C.byref(n))
        if rc:
            raise capi.LoraHipError(rc)
        if not n.value:
            return np.zeros((0, 3), np.uint64)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), (n.value, 3)).copy()

    def preload(self, start, idx, val):
        """lora_demod_core_preload: up-path results (offset 0) of the windows starting at framed-stream
        index start + k*N, produced ahead of time (shard.sharded_detect_sweep)."""
        idx = np.ascontiguousarray(idx, np.uint16); val = np.ascontiguousarray(val, np.float32)
        assert len(idx) == len(val)
        rc = capi.lib().lora_demod_core_preload(self._h, int(start), len(idx), idx.ctypes.data, val.ctypes.data)
        if rc:
            raise capi.LoraHipError(rc)

    def work(self, samples):
        """Feed complex64 samples; returns the PDUs completed by this call as u16 arrays.  A torch cuda tensor
        (e.g. the channeliser's output) is read where it lies in HBM (lora_demod_core_work_device)."""
        self._pending = _EMPTY
        if capi._is_torch(samples) and samples.is_cuda:
            import torch
            assert samples.dtype == torch.complex64 and samples.is_contiguous()
            cur = torch.cuda.current_stream(samples.device)
            if cur.cuda_stream:
                rc = capi.lib().lora_demod_core_set_stream(self._h, C.c_void_p(cur.cuda_stream))
            else:       # torch's default stream is the null stream, which the C ABI cannot name: wait for the producer
                cur.synchronize()
                rc = capi.lib().lora_demod_core_set_stream(self._h, None)
            if rc:
                raise capi.LoraHipError(rc)
            rc = capi.lib().lora_demod_core_work_device(self._h, samples.data_ptr(), samples.numel(), self._cb, None)
            self._fetch_pdus()            # PDUs completed before an error are still reported
            if rc:
                raise capi.LoraHipError(rc, capi.lib().lora_demod_core_last_error(self._h).decode())
            return self._pending
        x = np.ascontiguousarray(samples, np.complex64)
        # host samples run on the block's own stream: a torch stream bound by an earlier device call may be gone by now
        rc = capi.lib().lora_demod_core_set_stream(self._h, None)
        if rc:
            raise capi.LoraHipError(rc)
        rc = capi.lib().lora_demod_core_work(self._h, x.ctypes.data, len(x), self._cb, None)
        self._fetch_pdus()
        if rc:
            raise capi.LoraHipError(rc, capi.lib().lora_demod_core_last_error(self._h).decode())
        return self._pending

    @property
    def stats(self):
        L = capi.lib()
        return dict(calls=int(L.lora_demod_core_calls(self._h)), windows=int(L.lora_demod_core_windows(self._h)),
                    launches=int(L.lora_demod_core_launches(self._h)), state=int(L.lora_demod_core_state(self._h)),
                    plans=int(L.lora_demod_core_plans(self._h)), replans=int(L.lora_demod_core_replans(self._h)),
                    on_demand=int(L.lora_demod_core_on_demand(self._h)), rollbacks=int(L.lora_demod_core_rollbacks(self._h)))
