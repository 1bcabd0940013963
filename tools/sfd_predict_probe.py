"""How often does an SFD window's bin differ from the offset-0 window's bin at the same place turned by F*off bins?
(A chirp offset only rotates the spectrum; the two evaluations differ in rounding only.)  Developer probe.
  python tools/sfd_predict_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import gr_lora_amd as g
from gr_lora_amd import synth
import oracle, cases

def probe(sf, F, cap, name):
    capi = g.capi
    n = 1 << sf; s = n // 16; M = n * F
    h = capi.HipDemod(sf, F, cases.BETA)
    h.stage_iq(cap)
    cnt = (len(cap) - n) // s + 1
    a_idx, a_val = h.batch_strided(None, 0, s, cnt, chirp_off=0, chirp_off_step=0, kind=capi.WIN_SFD)
    tot = bad = 0
    for ph in range(16):
        # attempt phase ph: the window at position p uses the offset (p - ph*s) mod n
        off0 = (-ph * s) % n
        b_idx, b_val = h.batch_strided(None, 0, s, cnt, chirp_off=off0, chirp_off_step=s, kind=capi.WIN_SFD)
        off = (off0 + np.arange(cnt, dtype=np.int64) * s) % n
        pred = (a_idx.astype(np.int64) + F * off) % M
        d = pred != b_idx
        tot += cnt; bad += int(d.sum())
    print(f"{name} SF{sf} F{F}: {bad} of {tot} windows differ from the turned offset-0 bin ({bad / tot:.2e})", flush=True)
    h.close()

for sf, F in ((8, 2), (9, 2), (10, 1), (12, 2), (7, 4)):
    n = 1 << sf
    clean = synth.capture(sf, 400 * n, symbols_per_packet=64, seed=5)
    probe(sf, F, clean, "clean dense packets")
    rng = np.random.default_rng(sf)
    noisy = oracle.awgn(clean, 10.0 ** (-0 / 10.0), 7, cases.squelch_gain(sf, F, 0.0))
    probe(sf, F, noisy, "0 dB AWGN")
