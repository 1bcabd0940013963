"""Independent numpy/scipy double-precision recomputation of the oracle's L1 quantities.

TEST INFRASTRUCTURE ONLY.  Nothing here calls the C oracle: tables, window and spectra are
recomputed from the formulas with numpy.fft and
scipy.signal.windows.kaiser so that a slip in oracle/lora_oracle.c shows up as a disagreement.
"""
import numpy as np
from scipy.signal.windows import kaiser as _scipy_kaiser


def chirp_tables_np(sf):
    """demod_impl.cc:92-101 with the float `phase` / double `accumulator` mix made explicit."""
    n = 1 << sf
    phase = np.float32(-np.pi)
    acc = 0.0
    inc = (2 * np.pi) / n
    accs = np.empty(2 * n, np.float64)
    for i in range(2 * n):
        acc += float(phase)
        accs[i] = acc
        phase = np.float32(float(phase) + inc)
    up = (np.cos(accs) + 1j * np.sin(accs)).astype(np.complex64)
    down = (np.cos(accs) - 1j * np.sin(accs)).astype(np.complex64)
    return up, down


def kaiser_np(ntaps, beta):
    return _scipy_kaiser(ntaps, beta, sym=True).astype(np.float32)


def window_spectrum_np(iq, pos, chirp_off, kind, sf, fft_factor, beta=25.0, tables=None):
    """|X|^2 (float64, M bins) of one window, all arithmetic in double."""
    n = 1 << sf
    up, down = tables if tables is not None else chirp_tables_np(sf)
    x = np.asarray(iq[pos:pos + n], np.complex128)
    if kind == 0:
        x = x * down[chirp_off:chirp_off + n].astype(np.complex128) * kaiser_np(n, beta).astype(np.float64)
    else:
        x = x * up[chirp_off:chirp_off + n].astype(np.complex128)
    X = np.fft.fft(x, n * fft_factor)
    return X.real ** 2 + X.imag ** 2


def window_argmax_np(iq, pos, chirp_off, kind, sf, fft_factor, beta=25.0, tables=None):
    """(idx, max, relative top-2 margin) in double precision; lowest index wins ties."""
    p = window_spectrum_np(iq, pos, chirp_off, kind, sf, fft_factor, beta, tables)
    idx = int(np.argmax(p))          # numpy returns the first maximum
    top = float(p[idx])
    if not top > 0.0:
        return idx, top, 1.0
    q = p.copy(); q[idx] = -1.0
    second = float(q.max())
    return idx, top, (top - max(second, 0.0)) / top
