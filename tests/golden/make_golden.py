#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/.

The reference holds no golden vectors for the demod path (its QA files are empty) and cannot be
built or imported in this image, so these fixtures are produced by the CPU oracle
(oracle/lora_oracle.c) and are "parity unpinned" with respect to the real gr-lora binary.
They pin the oracle against regressions and give the HIP path fixed expected outputs.

  python tests/golden/make_golden.py            # rewrites tests/golden/*.npz and index.json

Per case (see tests/cases.py:GOLDEN_CASES) the .npz holds only data:
  tx_<p>                      transmitted symbols of packet p
  pdu_syms, pdu_len           expected PDUs (concatenated symbols + lengths)
  tr_pos/state/chirp_off/idx/consumed/val/n_sfd   per-call trace of general_work
  l1_pos/off/kind/idx/val/margin                  every window evaluated (L1 known answers)
  capture (only when commit_iq) complex64 IQ
index.json: parameters, capture sha256/length/gain, call and FFT counts, table known answers.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
import cases   # noqa: E402


def table_kats():
    out = {}
    for sf in range(7, 13):
        up, down = oracle.chirp_tables(sf)
        w = oracle.kaiser(1 << sf, cases.BETA)
        out[str(sf)] = dict(
            up_sha256=cases.sha256(up), down_sha256=cases.sha256(down), window_sha256=cases.sha256(w),
            up_head=[[float(z.real), float(z.imag)] for z in up[:8]],
            up_tail=[[float(z.real), float(z.imag)] for z in up[-8:]],
            window_head=[float(x) for x in w[:8]], window_sum=float(w.astype(np.float64).sum()))
    return out


def main():
    oracle.build()
    index = {"beta": cases.BETA, "sync_word": cases.SYNC_WORD, "tables": table_kats(), "cases": {}}
    for name, p in cases.GOLDEN_CASES.items():
        cap, tx, gain = cases.case_capture(p)
        d = oracle.Demod(p["sf"], bool(p["ldr"]), cases.BETA, p["fft_factor"])
        pdus, trace, n_calls, n_ffts = d.stream(cap, want_trace=True)
        n = 1 << p["sf"]
        pos, off, kind = cases.l1_windows_from_trace(trace, n)
        idx, val, margin = d.window_argmax(cases.gr_framed(cap, p["sf"]), pos, off, kind)
        # the L1 replay must reproduce the trace's own up-path values
        up_rows = np.flatnonzero(kind == oracle.KIND_UP)
        assert np.array_equal(idx[up_rows], trace["idx"]) and np.array_equal(val[up_rows], trace["val"])
        arrays = dict(
            pdu_syms=np.concatenate(pdus).astype(np.uint16) if pdus else np.zeros(0, np.uint16),
            pdu_len=np.array([len(x) for x in pdus], np.uint32),
            l1_pos=pos, l1_off=off, l1_kind=kind, l1_idx=idx, l1_val=val, l1_margin=margin)
        for f in trace.dtype.names:
            arrays["tr_" + f] = np.ascontiguousarray(trace[f])
        for i, s in enumerate(tx):
            arrays[f"tx_{i}"] = s
        if p["commit_iq"]:
            arrays["capture"] = cap
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
        index["cases"][name] = dict(p, capture_sha256=cases.sha256(cap), capture_len=int(len(cap)),
                                    gain=gain, n_calls=int(n_calls), n_ffts=int(n_ffts),
                                    n_pdus=len(pdus), n_windows=int(len(pos)),
                                    min_margin=float(margin.min()) if len(margin) else 1.0)
        print(f"{name}: {len(pdus)} PDUs, {n_calls} calls, {n_ffts} FFTs, {len(pos)} windows, "
              f"min margin {index['cases'][name]['min_margin']:.3g}")
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
