"""ctypes binding of the C ABI declared in include/lora_hip_demod.h.

Product code.  Loads gr-lora_amd/liblora_hip_demod.so (built in-tree by csrc/Makefile for gfx950)
and fails loudly when it is missing: there is no CPU fallback and nothing here touches oracle/.
PyTorch is used only as plumbing (device buffers, streams); signatures stay plain pointers.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LORA_HIP_LIB") or os.path.join(_HERE, "liblora_hip_demod.so")

OK, EINVAL, EUNSUPPORTED, ENOMEM, EHIP, ENODEVICE = 0, -1, -2, -3, -4, -5
NOALLOC_MAX_HANDLES = 16      # include/lora_hip_demod.h: LORA_HIP_NOALLOC_MAX_HANDLES
WIN_UP, WIN_SFD = 0, 1
IQ_ON_DEVICE, OUT_ON_DEVICE, DESC_ON_DEVICE, ASYNC, IQ_STAGED, OUT_PINNED, NO_VAL, LANE1, OUT_DIRECT = 1, 2, 4, 8, 16, 32, 64, 128, 256

WIN_DESC_DTYPE = np.dtype([("pos", "<u8"), ("chirp_off", "<u2"), ("kind", "u1"), ("reserved", "u1", (5,))])
assert WIN_DESC_DTYPE.itemsize == 16
# lora_win_run, include/lora_hip_demod.h
WIN_RUN_DTYPE = np.dtype([("pos0", "<u8"), ("stride", "<u4"), ("count", "<u4"), ("chirp_off", "<u2"),
                          ("chirp_off_step", "<u2"), ("kind", "u1"), ("reserved", "u1", (3,))])
assert WIN_RUN_DTYPE.itemsize == 24

# every symbol include/lora_hip_demod.h declares
EXPORTS = (
    "lora_hip_strerror", "lora_hip_last_hip_error", "lora_hip_make_chirp_tables", "lora_hip_make_kaiser",
    "lora_hip_demod_create", "lora_hip_demod_destroy", "lora_hip_demod_set_stream", "lora_hip_demod_batch",
    "lora_hip_demod_batch_strided", "lora_hip_demod_batch_strided_noalloc", "lora_hip_demod_batch_runs", "lora_hip_demod_sync",
    "lora_hip_demod_mark", "lora_hip_demod_wait",
    "lora_hip_host_alloc", "lora_hip_host_free", "lora_hip_demod_sf", "lora_hip_demod_fft_factor",
    "lora_hip_demod_kernel_name", "lora_hip_demod_stage_iq", "lora_hip_demod_stage_iq2",
    "lora_hip_demod_stage_iq2_dev", "lora_hip_demod_fetch", "lora_demod_core_work_device", "lora_demod_core_set_stream",
    # include/lora_demod_core.h
    "lora_demod_core_create", "lora_demod_core_destroy", "lora_demod_core_work", "lora_demod_core_set_trace",
    "lora_demod_core_calls", "lora_demod_core_windows", "lora_demod_core_launches", "lora_demod_core_state",
    "lora_demod_core_last_error", "lora_demod_core_preload", "lora_demod_core_reset", "lora_demod_core_set_plan", "lora_demod_core_seek", "lora_demod_core_idle_log", "lora_demod_core_set_collect", "lora_demod_core_collected",
    "lora_demod_core_plans", "lora_demod_core_replans", "lora_demod_core_on_demand", "lora_demod_core_rollbacks",
    "lora_demod_core_rule_hits",
    # include/lora_hip_chan.h
    "lora_hip_chan_design_taps", "lora_hip_chan_create", "lora_hip_chan_destroy", "lora_hip_chan_set_stream",
    "lora_hip_chan_last_hip_error", "lora_hip_chan_reset", "lora_hip_chan_output_count", "lora_hip_chan_work",
    "lora_hip_chan_taps_per_filter", "lora_hip_chan_dec_rate", "lora_hip_chan_flt_rate", "lora_hip_chan_kernel_name",
    "lora_hip_chan_force_general", "lora_hip_chan_piece_limit", "lora_hip_chan_phase_inc_effective", "lora_hip_chan_seek",
    # include/lora_decode.h
    "lora_decode_max_bytes", "lora_decode_pdu",
)

PDU_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint16), C.c_size_t)
CORE_TRACE_DTYPE = np.dtype([("pos", "<u8"), ("state", "<u2"), ("chirp_off", "<u2"), ("idx", "<u2"),
                             ("consumed", "<u2"), ("val", "<f4"), ("n_sfd", "<u4")], align=True)


class LoraHipError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().lora_hip_strerror(code).decode()
        super().__init__(f"lora_hip error {code}: {msg}" + (f" ({detail})" if detail else ""))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C gr-lora_amd/csrc` (hipcc, gfx950). "
            "The demod hot path has no CPU fallback.")
    # PyTorch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own
    # the GPU.  Import torch first so that this library binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u, i, f, d, sz, u64 = C.c_void_p, C.c_uint, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64
    L.lora_hip_strerror.argtypes = [i]; L.lora_hip_strerror.restype = C.c_char_p
    L.lora_hip_last_hip_error.argtypes = [vp]; L.lora_hip_last_hip_error.restype = C.c_char_p
    L.lora_hip_make_chirp_tables.argtypes = [u, vp, vp]; L.lora_hip_make_chirp_tables.restype = i
    L.lora_hip_make_kaiser.argtypes = [u, d, vp]; L.lora_hip_make_kaiser.restype = i
    L.lora_hip_demod_create.argtypes = [C.POINTER(vp), i, u, u, f, vp, vp, vp]; L.lora_hip_demod_create.restype = i
    L.lora_hip_demod_destroy.argtypes = [vp]; L.lora_hip_demod_destroy.restype = None
    L.lora_hip_demod_set_stream.argtypes = [vp, vp]; L.lora_hip_demod_set_stream.restype = i
    L.lora_hip_demod_batch.argtypes = [vp, vp, sz, vp, sz, vp, vp, u]; L.lora_hip_demod_batch.restype = i
    L.lora_hip_demod_batch_strided.argtypes = [vp, vp, sz, u64, u64, sz, u, u, u, vp, vp, u]
    L.lora_hip_demod_batch_strided.restype = i
    L.lora_hip_demod_batch_strided_noalloc.argtypes = [vp, vp, sz, u64, u64, sz, u, u, u, vp, vp]
    L.lora_hip_demod_batch_strided_noalloc.restype = i
    L.lora_hip_demod_sf.argtypes = [vp]; L.lora_hip_demod_sf.restype = u
    L.lora_hip_demod_fft_factor.argtypes = [vp]; L.lora_hip_demod_fft_factor.restype = u
    L.lora_hip_demod_kernel_name.argtypes = [vp]; L.lora_hip_demod_kernel_name.restype = C.c_char_p
    L.lora_hip_demod_stage_iq.argtypes = [vp, vp, sz]; L.lora_hip_demod_stage_iq.restype = i
    L.lora_hip_demod_stage_iq2.argtypes = [vp, vp, sz, vp, sz]; L.lora_hip_demod_stage_iq2.restype = i
    L.lora_hip_demod_stage_iq2_dev.argtypes = [vp, vp, sz, vp, sz]; L.lora_hip_demod_stage_iq2_dev.restype = i
    L.lora_hip_demod_fetch.argtypes = [vp, vp, sz, vp]; L.lora_hip_demod_fetch.restype = i
    L.lora_demod_core_work_device.argtypes = [vp, vp, sz, PDU_CB, vp]; L.lora_demod_core_work_device.restype = i
    L.lora_demod_core_set_stream.argtypes = [vp, vp]; L.lora_demod_core_set_stream.restype = i
    L.lora_demod_core_create.argtypes = [C.POINTER(vp), i, u, i, f, u]; L.lora_demod_core_create.restype = i
    L.lora_demod_core_destroy.argtypes = [vp]; L.lora_demod_core_destroy.restype = None
    L.lora_demod_core_work.argtypes = [vp, vp, sz, PDU_CB, vp]; L.lora_demod_core_work.restype = i
    L.lora_demod_core_set_trace.argtypes = [vp, vp, sz]; L.lora_demod_core_set_trace.restype = i
    L.lora_demod_core_set_plan.argtypes = [vp, i]; L.lora_demod_core_set_plan.restype = i
    L.lora_demod_core_seek.argtypes = [vp, u64]; L.lora_demod_core_seek.restype = i
    L.lora_demod_core_idle_log.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]; L.lora_demod_core_idle_log.restype = i
    L.lora_demod_core_set_collect.argtypes = [vp, i]; L.lora_demod_core_set_collect.restype = i
    L.lora_demod_core_collected.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]; L.lora_demod_core_collected.restype = i
    L.lora_hip_demod_batch_runs.argtypes = [vp, vp, sz, vp, sz, vp, vp, u]; L.lora_hip_demod_batch_runs.restype = i
    L.lora_hip_demod_sync.argtypes = [vp, i]; L.lora_hip_demod_sync.restype = i
    L.lora_hip_demod_mark.argtypes = [vp, i, i]; L.lora_hip_demod_mark.restype = i
    L.lora_hip_demod_wait.argtypes = [vp, i]; L.lora_hip_demod_wait.restype = i
    L.lora_hip_host_alloc.argtypes = [C.POINTER(vp), sz]; L.lora_hip_host_alloc.restype = i
    L.lora_hip_host_free.argtypes = [vp]; L.lora_hip_host_free.restype = None
    for nme in ("lora_demod_core_calls", "lora_demod_core_windows", "lora_demod_core_launches",
                "lora_demod_core_plans", "lora_demod_core_replans", "lora_demod_core_on_demand", "lora_demod_core_rollbacks"):
        getattr(L, nme).argtypes = [vp]; getattr(L, nme).restype = u64
    L.lora_demod_core_state.argtypes = [vp]; L.lora_demod_core_state.restype = i
    L.lora_demod_core_rule_hits.argtypes = [vp, vp, f, sz, vp, i]; L.lora_demod_core_rule_hits.restype = sz
    L.lora_demod_core_last_error.argtypes = [vp]; L.lora_demod_core_last_error.restype = C.c_char_p
    L.lora_demod_core_reset.argtypes = [vp]; L.lora_demod_core_reset.restype = i
    L.lora_demod_core_preload.argtypes = [vp, u64, sz, vp, vp]; L.lora_demod_core_preload.restype = i
    L.lora_hip_chan_piece_limit.argtypes = [u]; L.lora_hip_chan_piece_limit.restype = sz
    L.lora_hip_chan_design_taps.argtypes = [d, u, d, vp, sz, C.POINTER(sz)]; L.lora_hip_chan_design_taps.restype = i
    L.lora_hip_chan_create.argtypes = [C.POINTER(vp), i, d, f, vp, sz, u]; L.lora_hip_chan_create.restype = i
    L.lora_hip_chan_destroy.argtypes = [vp]; L.lora_hip_chan_destroy.restype = None
    L.lora_hip_chan_set_stream.argtypes = [vp, vp]; L.lora_hip_chan_set_stream.restype = i
    L.lora_hip_chan_last_hip_error.argtypes = [vp]; L.lora_hip_chan_last_hip_error.restype = C.c_char_p
    L.lora_hip_chan_reset.argtypes = [vp]; L.lora_hip_chan_reset.restype = i
    L.lora_hip_chan_output_count.argtypes = [vp, sz]; L.lora_hip_chan_output_count.restype = sz
    L.lora_hip_chan_work.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), u]; L.lora_hip_chan_work.restype = i
    for nme in ("lora_hip_chan_taps_per_filter", "lora_hip_chan_dec_rate"):
        getattr(L, nme).argtypes = [vp]; getattr(L, nme).restype = u
    L.lora_hip_chan_flt_rate.argtypes = [vp]; L.lora_hip_chan_flt_rate.restype = f
    L.lora_hip_chan_kernel_name.argtypes = [vp]; L.lora_hip_chan_kernel_name.restype = C.c_char_p
    L.lora_hip_chan_force_general.argtypes = [vp, i]; L.lora_hip_chan_force_general.restype = i
    L.lora_hip_chan_phase_inc_effective.argtypes = [vp]; L.lora_hip_chan_phase_inc_effective.restype = d
    L.lora_hip_chan_seek.argtypes = [vp, u64, vp, u]; L.lora_hip_chan_seek.restype = i
    L.lora_decode_max_bytes.argtypes = [sz]; L.lora_decode_max_bytes.restype = sz
    L.lora_decode_pdu.argtypes = [u, u, i, vp, sz, vp, sz, vp, sz, C.POINTER(sz)]; L.lora_decode_pdu.restype = i
    _lib = L
    return L


def make_chirp_tables(sf):
    n = 1 << sf
    up = np.empty(2 * n, np.complex64); down = np.empty(2 * n, np.complex64)
    rc = lib().lora_hip_make_chirp_tables(sf, up.ctypes.data, down.ctypes.data)
    if rc:
        raise LoraHipError(rc)
    return up, down


def make_kaiser(ntaps, beta):
    w = np.empty(ntaps, np.float32)
    rc = lib().lora_hip_make_kaiser(ntaps, float(beta), w.ctypes.data)
    if rc:
        raise LoraHipError(rc)
    return w


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    """(address, on_device) of a numpy array or a torch tensor (contiguous)."""
    if x is None:
        return None, False
    if _is_torch(x):
        assert x.is_contiguous()
        return x.data_ptr(), x.is_cuda
    assert x.flags["C_CONTIGUOUS"]
    return x.ctypes.data, False


class HipDemod:
    """One lora_hip_demod_t handle: the GPU state of one demod block instance."""

    def __init__(self, sf, fft_factor, beta=25.0, device=0, window=None, upchirp=None, downchirp=None):
        self.sf, self.fft_factor, self.N = sf, fft_factor, 1 << sf
        self._staged_n = 0
        self._h = C.c_void_p()
        self._destroy = lib().lora_hip_demod_destroy      # bound now: module globals are gone at interpreter exit
        args = [None, None, None]
        if window is not None:
            self._tabs = (np.ascontiguousarray(window, np.float32), np.ascontiguousarray(upchirp, np.complex64),
                          np.ascontiguousarray(downchirp, np.complex64))
            args = [t.ctypes.data for t in self._tabs]
        rc = lib().lora_hip_demod_create(C.byref(self._h), device, sf, fft_factor, beta, *args)
        if rc:
            self._h = None
            raise LoraHipError(rc)

    def close(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def kernel_name(self):
        return lib().lora_hip_demod_kernel_name(self._h).decode()

    def set_stream(self, stream_ptr):
        rc = lib().lora_hip_demod_set_stream(self._h, C.c_void_p(stream_ptr or 0))
        if rc:
            raise LoraHipError(rc)

    def _check(self, rc):
        if rc:
            raise LoraHipError(rc, lib().lora_hip_last_hip_error(self._h).decode())

    def _outputs(self, n, out_idx, out_val):
        if out_idx is None:
            out_idx = np.empty(n, np.uint16); out_val = np.empty(n, np.float32)
        return out_idx, out_val

    def batch(self, iq, desc, out_idx=None, out_val=None, async_=False):
        """lora_hip_demod_batch.  iq/desc/outputs: numpy (host) or torch cuda tensors (device)."""
        n_win = len(desc) if not _is_torch(desc) else desc.shape[0]
        out_idx, out_val = self._outputs(n_win, out_idx, out_val)
        pi, di = _ptr(iq); pd, dd = _ptr(desc); po, do = _ptr(out_idx); pv, dv = _ptr(out_val)
        assert do == dv
        flags = (IQ_ON_DEVICE if di else 0) | (DESC_ON_DEVICE if dd else 0) | (OUT_ON_DEVICE if do else 0) | (ASYNC if async_ else 0)
        n_samples = iq.numel() if _is_torch(iq) else len(iq)
        if _is_torch(iq) and iq.dtype.is_floating_point and not iq.dtype.is_complex:
            n_samples //= 2
        self._check(lib().lora_hip_demod_batch(self._h, pi, n_samples, pd, n_win, po, pv, flags))
        return out_idx, out_val

    def stage_iq(self, iq):
        """lora_hip_demod_stage_iq: copy a host capture into the handle's device buffer once; later
        calls pass iq=None and read it there."""
        x = np.ascontiguousarray(iq, np.complex64)
        self._check(lib().lora_hip_demod_stage_iq(self._h, x.ctypes.data, len(x)))
        self._staged_n = len(x)

    def batch_strided(self, iq, pos0, stride, count, chirp_off=0, chirp_off_step=0, kind=WIN_UP,
                      out_idx=None, out_val=None, async_=False):
        out_idx, out_val = self._outputs(count, out_idx, out_val)
        pi, di = _ptr(iq); po, do = _ptr(out_idx); pv, dv = _ptr(out_val)
        assert do == dv
        flags = (IQ_ON_DEVICE if di else 0) | (OUT_ON_DEVICE if do else 0) | (ASYNC if async_ else 0)
        if iq is None:                        # the capture staged by stage_iq()
            flags |= IQ_STAGED
            n_samples = self._staged_n
        else:
            n_samples = iq.numel() if _is_torch(iq) else len(iq)
        if iq is not None and _is_torch(iq) and iq.dtype.is_floating_point and not iq.dtype.is_complex:
            n_samples //= 2
        self._check(lib().lora_hip_demod_batch_strided(self._h, pi, n_samples, pos0, stride, count, chirp_off,
                                                       chirp_off_step, kind, po, pv, flags))
        return out_idx, out_val

    def batch_runs(self, iq, runs, out_idx=None, out_val=None):
        """lora_hip_demod_batch_runs: runs = [(pos0, stride, count, chirp_off, chirp_off_step, kind), ...] evaluated in one
        launch, results back to back in run order."""
        r = np.zeros(len(runs), WIN_RUN_DTYPE)
        for k, u in enumerate(runs):
            r["pos0"][k], r["stride"][k], r["count"][k], r["chirp_off"][k], r["chirp_off_step"][k], r["kind"][k] = u
        count = int(r["count"].sum())
        out_idx, out_val = self._outputs(count, out_idx, out_val)
        pi, di = _ptr(iq); po, do = _ptr(out_idx); pv, dv = _ptr(out_val)
        assert do == dv
        flags = (IQ_ON_DEVICE if di else 0) | (OUT_ON_DEVICE if do else 0)
        if iq is None:
            flags |= IQ_STAGED
            n_samples = self._staged_n
        else:
            n_samples = iq.numel() if _is_torch(iq) else len(iq)
        self._check(lib().lora_hip_demod_batch_runs(self._h, pi, n_samples, r.ctypes.data, len(r), po, pv, flags))
        return out_idx, out_val

    def batch_strided_noalloc(self, iq, pos0, stride, count, out_idx, out_val, chirp_off=0, chirp_off_step=0,
                              kind=WIN_UP):
        """lora_hip_demod_batch_strided_noalloc: device tensors only, enqueues on the handle's stream and
        returns; the caller synchronises."""
        pi, di = _ptr(iq); po, do = _ptr(out_idx); pv, dv = _ptr(out_val)
        assert di and do and dv, "the no-allocation entry point takes device buffers only"
        n_samples = iq.numel()
        if iq.dtype.is_floating_point and not iq.dtype.is_complex:
            n_samples //= 2
        self._check(lib().lora_hip_demod_batch_strided_noalloc(self._h, pi, n_samples, pos0, stride, count, chirp_off,
                                                               chirp_off_step, kind, po, pv))


def make_desc(pos, chirp_off, kind):
    pos = np.asarray(pos, np.uint64)
    d = np.zeros(len(pos), WIN_DESC_DTYPE)
    d["pos"] = pos
    d["chirp_off"] = np.broadcast_to(np.asarray(chirp_off, np.uint16), pos.shape)
    d["kind"] = np.broadcast_to(np.asarray(kind, np.uint8), pos.shape)
    return d
