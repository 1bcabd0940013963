#!/usr/bin/env python3
"""Randomised differential test of the whole block (GPU windows + host state machine) against the oracle.
Random SF / fft_factor / low_data_rate, packet lengths, gaps (arbitrary sample counts), AWGN levels around
the squelch threshold, amplitude scaling, and chunked feeding.  Uses the L2 bar of tests/test_gpu_l2_parity.py:
identical call sequence and PDUs, or a divergence that starts at a window whose top-2 margin is below 1e-4.

  python tools/fuzz_l2.py [seconds] [seed] [any]      # "any": draw fft_factor from ANY_FACTORS too
  LORA_CORE_PLAN=2 LORA_CORE_STRETCH=2048 LORA_FUZZ_PACKETS=24 python tools/fuzz_l2.py 300 7
                                                      # every input planned, in stretches of 128 symbols: the pipelined block
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gr_lora_amd as g
import oracle, cases
from test_gpu_l2_parity import compare_with_oracle


ANY_FACTORS = [1, 2, 4, 3, 3, 5, 6, 7, 8, 12, 15]      # beside the specialised 1/2/4: the any-factor kernels


def random_case(rng, factors=(1, 2, 2, 4)):
    sf = int(rng.choice([6, 7, 7, 8, 8, 9, 10, 11, 12]))
    f = int(rng.choice(list(factors)))
    ldr = int(rng.integers(0, 2))
    n = 1 << sf
    n_pk = int(rng.integers(1, 1 + int(os.environ.get("LORA_FUZZ_PACKETS", "4"))))     # more packets: inputs long enough for several stretches
    parts = []
    for k in range(n_pk):
        parts.append(np.zeros(int(rng.integers(0, 6 * n)), np.complex64))
        ns = int(rng.integers(1, 40 if sf < 11 else 14))
        parts.append(oracle.modulate(sf, oracle.random_symbols(sf, ns, int(rng.integers(1 << 30))), cases.SYNC_WORD))
        if rng.random() < 0.2:                     # a truncated packet
            parts[-1] = parts[-1][: int(rng.integers(n, len(parts[-1])))]
    parts.append(np.zeros(4 * n + int(rng.integers(0, n)), np.complex64))
    cap = np.concatenate(parts)
    mode = rng.random()
    if mode < 0.6:
        snr = float(rng.uniform(-15, 15))
        gain = cases.squelch_gain(sf, f, snr) * float(rng.uniform(0.5, 2.0))     # around the squelch threshold
        cap = oracle.awgn(cap, 10.0 ** (-snr / 10.0), int(rng.integers(1 << 30)), gain)
    elif mode < 0.8:
        cap = (cap * np.float32(rng.uniform(0.001, 3.0))).astype(np.complex64)
    chunks = None
    big = int(os.environ.get("LORA_FUZZ_CHUNK_SYMS", "0"))        # > 0: chunks of up to this many symbols (default-mode planning of
    if big and rng.random() < 0.7:                                # dense streams sets in at 192 symbols per call)
        chunks = [int(c) for c in rng.integers(64 * n, big * n, int(rng.integers(1, 30)))]
    elif rng.random() < 0.4:
        chunks = [int(c) for c in rng.integers(1, 3 * n, int(rng.integers(1, 30)))]
    return sf, f, ldr, cap, chunks


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    factors = ANY_FACTORS if len(sys.argv) > 3 and sys.argv[3] == "any" else (1, 2, 2, 4)
    rng = np.random.default_rng(seed)
    t0, n_case, n_same, n_div, n_pdus, n_calls = time.time(), 0, 0, 0, 0, 0
    max_cases = int(os.environ.get("LORA_FUZZ_CASES", "0"))       # > 0: stop after this many cases (same corpus for an A/B)
    while time.time() - t0 < budget and (max_cases == 0 or n_case < max_cases):
        sf, f, ldr, cap, chunks = random_case(rng, factors)
        try:
            r = compare_with_oracle(g, sf, f, ldr, cap, chunks=chunks, name=f"case {n_case} sf{sf} f{f} ldr{ldr}")
        except AssertionError as e:
            np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{n_case}.npy"), cap)
            print(f"FAIL case {n_case}: sf={sf} f={f} ldr={ldr} chunks={chunks} len={len(cap)}: {e}", flush=True)
            raise SystemExit(1)
        n_case += 1; n_pdus += r["pdus"]; n_calls += r["calls"]
        if r["same_path"]:
            n_same += 1
        else:
            n_div += 1
        if n_case % 50 == 0:
            print(f"{n_case} cases, {n_same} identical call paths, {n_div} sub-margin divergences, {n_pdus} PDUs, {n_calls} calls", flush=True)
    print(f"fuzz seed {seed}: {n_case} cases in {time.time() - t0:.0f} s: {n_same} identical call paths, "
          f"{n_div} divergences all attributed to windows below the 1e-4 margin, {n_pdus} PDUs, {n_calls} calls: OK", flush=True)


if __name__ == "__main__":
    main()
