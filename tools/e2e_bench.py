"""End-to-end and SFD-sweep measurements that are reported beside (never as) the headline value."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gr_lora_amd as g
from importlib import import_module
synth = import_module("gr-lora_amd.synth")

def sfd_sweep(sf, f, total_bytes=256 << 20, reps=5):
    n = 1 << sf
    ns = total_bytes // 8
    dev = torch.device("cuda:0")
    iq = torch.randn(ns, 2, device=dev, dtype=torch.float32)
    count = (ns - n) // (n // 16) + 1
    oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
    h = g.HipDemod(sf, f)
    s = torch.cuda.Stream(); h.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        h.batch_strided(iq, 0, n // 16, count, 0, n // 16, g.capi.WIN_SFD, out_idx=oi, out_val=ov, async_=True); s.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            h.batch_strided(iq, 0, n // 16, count, 0, n // 16, g.capi.WIN_SFD, out_idx=oi, out_val=ov, async_=True)
        e1.record(s); s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"SFD sweep SF{sf} F{f}: {count} overlapped windows (stride N/16) {ms:.3f} ms -> {count/ms/1e3:.1f} Mwin/s "
          f"= {count/16/ms/1e3:.2f} M symbol-times/s", flush=True)
    h.close()

def single_capture_latency(sf=12, f=2, n_samples=1_250_000, reps=50):
    """BASELINE cfg 2 as one call: a 10 s capture resident in HBM, every stride-N window, host sees the results."""
    n = 1 << sf
    cap = synth.capture(sf, n_samples, symbols_per_packet=64, seed=2)
    iq = torch.from_numpy(cap).cuda()
    count = len(cap) // n
    h = g.HipDemod(sf, f)
    oi = torch.empty(count, dtype=torch.int16, device="cuda"); ov = torch.empty(count, dtype=torch.float32, device="cuda")
    for _ in range(3):
        h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov)
    t0 = time.perf_counter()
    for _ in range(reps):
        h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov)       # synchronous: returns when results are in HBM
    dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        h.batch_strided(iq, 0, n, count)                                  # results copied to host memory
    host = (time.perf_counter() - t0) / reps
    # the same call recorded 64 times into a HIP graph (LORA_HIP_ASYNC on the capture stream) and replayed
    st = torch.cuda.Stream(); h.set_stream(st.cuda_stream)
    h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov, async_=True); st.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        for _ in range(64):
            h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov, async_=True)
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    gr = (time.perf_counter() - t0) / 640
    print(f"single capture SF{sf} F{f}: {count} windows of a {n_samples}-sample capture in HBM: {dev*1e6:.0f} us per call "
          f"(results in HBM), {host*1e6:.0f} us with results in host memory, {gr*1e6:.1f} us per call inside a replayed "
          f"HIP graph of 64 calls", flush=True)
    h.close()

def e2e(sf, f, seconds_of_capture=40):
    cap = synth.capture(sf, int(125e3 * seconds_of_capture), symbols_per_packet=64, seed=3)
    blk = g.demod(sf, False, 25.0, f)
    blk.work(cap[: 64 << sf])            # warm up
    blk.close()
    blk = g.demod(sf, False, 25.0, f)
    t0 = time.perf_counter()
    pdus = blk.work(cap)
    dt = time.perf_counter() - t0
    st = blk.stats
    print(f"end-to-end demod.work SF{sf} F{f}: {len(cap)} samples from host memory, {len(pdus)} PDUs, {st['calls']} calls, "
          f"{st['windows']} GPU windows in {st['launches']} launches, {dt*1e3:.1f} ms -> {st['calls']/dt/1e6:.3f} Msym/s "
          f"({len(cap)*8/dt/1e9:.2f} GB/s of IQ incl. PCIe)", flush=True)
    blk.close()

def e2e_design_a(sf, f, seconds_of_capture=40):
    """16-phase detect sweep ahead of time (what N ranks would share), then the replay on one block."""
    shard = import_module("gr-lora_amd.shard")
    cap = synth.capture(sf, int(125e3 * seconds_of_capture), symbols_per_packet=64, seed=3)
    ref = g.demod(sf, False, 25.0, f); ref_pdus = ref.work(cap); ref.close()
    sweep = shard.HipSweeper(sf, f)
    t0 = time.perf_counter()
    runs = shard.sharded_detect_sweep(cap, sf, sweep)
    t1 = time.perf_counter()
    blk = g.demod(sf, False, 25.0, f)
    for start, idx, val in runs:
        blk.preload(start, idx, val)
    pdus = blk.work(cap)
    t2 = time.perf_counter()
    st = blk.stats
    same = [p.tolist() for p in pdus] == [p.tolist() for p in ref_pdus]
    nwin = sum(len(r[1]) for r in runs)
    print(f"design-A SF{sf} F{f}: detect sweep of {nwin} windows (16 phases) {1e3*(t1-t0):.1f} ms, replay {1e3*(t2-t1):.1f} ms "
          f"with {st['launches']} launches / {st['windows']} windows; PDUs identical to the sequential block: {same}", flush=True)
    blk.close(); sweep.close()

if __name__ == "__main__":
    single_capture_latency()
    for sf, f in ((12, 2), (10, 2), (8, 2)):
        sfd_sweep(sf, f)
    for sf, f in ((12, 2), (9, 2), (8, 2)):
        e2e(sf, f)
    for sf, f in ((12, 2), (9, 2)):
        e2e_design_a(sf, f)
