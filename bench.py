#!/usr/bin/env python3
"""Headline bench: demodulated symbols/s at SF12, fft_factor 2 (BASELINE.json `metric`).

  python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of the hot path (lora_hip_demod_batch_strided: dechirp, Kaiser window,
zero-padded 8192-point FFT, |X|^2 argmax) over every stride-N symbol window of the synthetic
capture set resident in HBM.  Workload = BASELINE.json configs[1]: the 10 s x 125 kS/s SF12
capture (1 250 000 samples, chirp packets from the mod layout), replicated R times per GPU so
that >= 1 GiB of IQ is resident (a single capture is only 305 windows, far too short to time).
Each rank owns its own replicas (independent captures shard, no data-path collective).
Prints ONE JSON line on rank 0.

Launched by torch.distributed.run (WORLD_SIZE set) every process is one rank.  Started plainly with --gpus N > 1
(WORLD_SIZE unset) this process becomes a parent that never touches the GPU: it starts N fresh rank processes with
RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set and relays rank 0's line.

Before the timed region every rank runs a fixed clock-settling pre-roll (--settle-s seconds of back-to-back launches,
reported as `settle_launches`): the device boosts for the first few launches, throttles, and settles after a few hundred
(profiles/r01_launch_trace.txt), so a short --warmup alone would time the transient.  Per-step times (HIP events on the
launch stream) are reported as min/median/max so that a transient inside the timed region would be visible.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SF, FFT_FACTOR, BETA = 12, 2, 25.0
CAPTURE_SAMPLES = 1_250_000          # 10 s at 125 kS/s
MIN_RESIDENT_BYTES = 1 << 30
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


_WORKER = """
import sys, numpy as np
sys.path.insert(0, %r)
import oracle
cap = np.load(sys.argv[1])
sec, calls, ffts, pdus = oracle.time_stream(%d, False, %r, %d, cap, reps=int(sys.argv[2]))
print(sec, ffts)
"""


def cpu_baseline(capture, budget_s=12.0):
    """The oracle's faithful restatement of general_work timed on the host cores (rank 0 only):
    one core (the reference block is single-threaded: fft_complex(..., 1), one scheduler thread), then
    one independent stream per core on up to 16 cores (the GNU Radio analogue of many channels)."""
    import subprocess
    import tempfile
    import numpy as np
    import oracle
    oracle.build()
    sec, calls, ffts, pdus = oracle.time_stream(SF, False, BETA, FFT_FACTOR, capture, reps=1)
    reps = max(1, int(budget_s / max(sec, 1e-3)))
    sec, calls, ffts, pdus = oracle.time_stream(SF, False, BETA, FFT_FACTOR, capture, reps=reps)
    out = {
        "value": round(ffts * reps / sec / 1e6, 6), "unit": "Msym/s", "cores": 1, "kind": "port",
        "unit_note": "counts FFT windows (every up-path window and every one of the 16 SFD windows of a sync call), the same unit as "
                     "the GPU's stride-N windows; general_work calls alone are the `calls` figure in `sample`",
        "sample": (f"oracle general_work loop (3 allocs/call, dead multiply, memset+memcpy, full 8192-pt radix-2 FFT, "
                   f"scalar argmax; own FFT, not FFTW/VOLK) over the same 1.25 M-sample SF12 capture x {reps} passes: "
                   f"{calls} calls / {ffts} FFT windows / {pdus} PDUs per pass, {sec:.1f} s; "
                   f"value counts every FFT window, calls alone = {calls * reps / sec / 1e6:.6f} Msym/s"),
    }
    # the same loop with the costs the reference pays for nothing taken out (SURVEY 8d "optimised CPU"): no per-call
    # allocation, no dead multiply, F pruned N-point four-step transforms (vectorised) instead of one zero-padded
    # M-point radix-2 transform, |X|^2 scan as a vector pass
    try:
        osec, ocalls, offts, opdus = oracle.time_stream(SF, False, BETA, FFT_FACTOR, capture, reps=1, optimised=True)
        oreps = max(1, int(budget_s / 2 / max(osec, 1e-3)))
        osec, ocalls, offts, opdus = oracle.time_stream(SF, False, BETA, FFT_FACTOR, capture, reps=oreps, optimised=True)
        out["optimised"] = {
            "value": round(offts * oreps / osec / 1e6, 6), "unit": "Msym/s", "cores": 1, "kind": "port",
            "sample": (f"same capture x {oreps} passes through the same state machine with the window evaluation rewritten "
                       f"(no allocation, no dead multiply, pruned four-step FFT, vector |X|^2 scan; own FFT, not FFTW/VOLK): "
                       f"{ocalls} calls / {offts} FFT windows / {opdus} PDUs per pass, {osec:.1f} s"),
        }
    except Exception as e:
        out["optimised"] = {"error": str(e)[:200]}
    # all cores of this box's share: separate processes that never touch the GPU
    ncores = max(1, min(16, os.cpu_count() or 1))
    try:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "cap.npy")
            np.save(path, capture)
            code = _WORKER % (ROOT, SF, BETA, FFT_FACTOR)
            wreps = max(1, reps // 2)
            t0 = time.perf_counter()
            procs = []
            try:
                for _ in range(ncores):
                    procs.append(subprocess.Popen([sys.executable, "-c", code, path, str(wreps)],
                                                  stdout=subprocess.PIPE, text=True))
                done = [p.communicate(timeout=300)[0].split() for p in procs]
            finally:
                for p in procs:                 # never leave a worker behind, whatever happened above
                    if p.poll() is None:
                        p.kill()
                    p.wait()
            wall = time.perf_counter() - t0
        if all(p.returncode == 0 for p in procs):
            total = sum(float(d[1]) for d in done) * wreps
            slowest = max(float(d[0]) for d in done)
            out["all_cores"] = {"value": round(total / slowest / 1e6, 6), "unit": "Msym/s", "cores": ncores,
                                "sample": f"{ncores} processes, one independent stream each, {wreps} passes, "
                                          f"slowest {slowest:.1f} s (wall incl. start-up {wall:.1f} s)"}
    except Exception as e:                      # the single-core figure is the contract; this one is extra
        out["all_cores"] = {"error": str(e)[:200]}
    return out


def kernel_header_sha256():
    """SHA-256 of the device code the counter files under profiles/ were measured on (tools/hbm_traffic.py and
    tools/valu_count.py store it): a counter figure is only passed on when the kernel source is still that one."""
    import hashlib
    with open(os.path.join(ROOT, "gr-lora_amd", "csrc", "lora_hip_kernels.hip.h"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def channeliser_leg(torch, g, dev, stream, total_bytes=1 << 30, reps=20):
    """Beside the headline (never part of `value`): the channeliser in front of the block at the
    reference flowgraph's setting -- 1 MS/s -> 250 kS/s, offset -250 kHz (examples/rx_usrp.grc:95,270,401) -- over
    1 GiB of complex64 resident in HBM, timed with HIP events on the launch stream."""
    import math
    n = total_bytes // 8
    x = torch.view_as_complex(torch.randn(n, 2, device=dev, dtype=torch.float32))
    ch = g.channelizer(2 * math.pi * (-250e3) / 1e6, 250e3 / 1e6, device=dev.index or 0)
    out = torch.empty(ch.output_count(n) + 8, dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()
    ch.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        for _ in range(5):
            ch.work(x, out=out, async_=True)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            y = ch.work(x, out=out, async_=True)
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = 8 * (n + len(y)) / (ms * 1e-3) / 1e9
    res = {"kernel": ch.kernel_name, "workload": "rate 0.25, offset -0.25 cycles/sample, 100 dB taps (91 per arm), 1 GiB in",
           "kernel_ms": round(ms, 4), "input_gsamp_per_s": round(n / ms / 1e6, 1),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_input_sample": 8 + 8 * 0.25}}
    ch.close()
    return res


def block_leg(torch, g, dev, n_samples=1 << 23):
    """Beside the headline (never part of `value`): the whole block -- GPU windows + the reference's state machine on the
    host (lora_demod_core, demod_impl.cc:149-450) -- over a dense SF8 capture resident in HBM (the case of
    tools/pipeline_bench.py without the channeliser): milliseconds and kernel launches per work() call."""
    import time as _t
    from importlib import import_module
    synth = import_module("gr-lora_amd.synth")
    cap = torch.from_numpy(synth.capture(8, n_samples, symbols_per_packet=64, seed=5)).to(dev)
    blk = g.demod(8, False, BETA, 2, device=dev.index or 0)
    best, pdus, st0 = None, 0, None
    for _ in range(4):
        blk.reset()
        st0 = blk.stats
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        pdus = len(blk.work(cap))
        dt = _t.perf_counter() - t0
        best = dt if best is None or dt < best else best
    st = blk.stats
    blk.close()
    return {"workload": f"SF8 F=2, {n_samples} samples in HBM, 64-symbol packets back to back", "pdus": pdus,
            "calls": st["calls"], "launches": st["launches"], "on_demand": st["on_demand"], "rollbacks": st["rollbacks"],
            "ms": round(best * 1e3, 3), "calls_per_s": round(st["calls"] / best, 0)}


def spawn_ranks(args):
    """--gpus N without a launcher: this parent (which has not touched the GPU and never does) starts N rank
    processes of this file and relays their output; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rcs = [p.wait() for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
            p.wait()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-s", type=float, default=0.5, help="seconds of untimed launches before --warmup (clock settling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: rendezvous, barrier and the JSON line only (CPU test of the launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    from importlib import import_module
    synth = import_module("gr-lora_amd.synth")
    cap = synth.capture(SF, CAPTURE_SAMPLES, symbols_per_packet=64, seed=2 + rank)

    # CPU baseline first, while this process has not initialised the GPU (it builds the oracle and starts worker
    # processes): rank 0 at N = 1 only
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.dry_run:
        cpu = cpu_baseline(cap)

    import torch
    if world > 1:
        import torch.distributed as dist
        # The data path has no collective (captures are independent): the process group only carries
        # the timing barrier and the max-over-ranks of two floats, so gloo on the host is enough and
        # RCCL/xGMI stay idle by design.
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout for the one JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    if args.dry_run:
        if world > 1:
            dist.barrier()
            t = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            assert float(t[0]) == world
        if rank == 0:
            print(json.dumps({"metric": "demodulated symbols/sec (Msym/s) at SF12 fft_factor=2", "value": None,
                              "unit": "Msym/s", "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    import gr_lora_amd as g
    ndev = torch.cuda.device_count()
    dev_index = local_rank if local_rank < ndev else local_rank % max(ndev, 1)   # rehearsal on fewer GPUs than ranks
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    n = 1 << SF
    replicas = -(-MIN_RESIDENT_BYTES // (cap.nbytes))
    iq = torch.from_numpy(cap).to(dev).repeat(replicas)           # complex64, resident in HBM
    torch.cuda.synchronize()                                      # built on torch's stream, read on `stream` below
    n_samples = iq.numel()
    count = n_samples // n                                        # stride-N windows per step
    out_idx = torch.empty(count, dtype=torch.int16, device=dev)
    out_val = torch.empty(count, dtype=torch.float32, device=dev)

    h = g.HipDemod(SF, FFT_FACTOR, BETA, device=dev_index)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)

    def step():
        h.batch_strided(iq, 0, n, count, 0, 0, g.capi.WIN_UP, out_idx=out_idx, out_val=out_val, async_=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # what a plain device copy of the same buffer reaches on this GPU (read + write bytes per second):
    # the practical ceiling beside the 8 TB/s specification.  Set-up work, done on every rank
    # before the warm-up steps, outside the timed region.
    raw = iq.view(torch.float32)
    dst = torch.empty_like(raw)
    with torch.cuda.stream(stream):
        dst.copy_(raw)
        c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
        c0.record(stream)
        for _ in range(20):
            dst.copy_(raw)
        c1.record(stream)
    stream.synchronize()
    copy_gbs = 2 * raw.numel() * 4 * 20 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del dst, raw

    # clock-settling pre-roll: back-to-back launches for --settle-s seconds, whatever --warmup says
    settle_launches = 0
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < args.settle_s:
        for _ in range(50):
            step()
        stream.synchronize()
        settle_launches += 50
    for _ in range(args.warmup):
        step()
    fence()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(stream)                   # HIP events on the launch stream: kernel time only
    for i in range(args.steps):
        step()
        evs[i + 1].record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = evs[0].elapsed_time(evs[-1]) / args.steps
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    step_stats = {"min": round(per_step[0], 4), "median": round(per_step[len(per_step) // 2], 4), "max": round(per_step[-1], 4)}

    if world > 1:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    # sanity: the preamble windows of the first packet must have been found (not a timing no-op)
    if int((out_val > 1.0).sum()) < count // 2:
        raise SystemExit("bench sanity check failed: the demod produced no signal windows")

    if rank == 0:
        value = world * count * args.steps / elapsed / 1e6
        bytes_per_sym = 8 * n + 8                                  # IQ read once + (idx,val) written
        achieved = count * bytes_per_sym / (kernel_ms * 1e-3) / 1e9
        # Counter figures are NOT measured in this run: they come from rocprofv3 --pmc passes recorded under profiles/
        # (a counter pass cannot run inside a timed bench).  Each file names the SHA-256 of the kernel source it was
        # measured on; when the source in this tree is another one, the figure is withheld (null) and the source field says so.
        sha = kernel_header_sha256()
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                j = json.load(f)
            same = j.get("kernel_header_sha256") == sha
            traffic_source = {"file": "profiles/hbm_traffic.json", "measured_in_this_run": False,
                              "kernel_header_sha256": j.get("kernel_header_sha256"), "matches_current_source": same}
            if same and j.get("kernel") == h.kernel_name and j.get("windows_per_launch") == count:
                traffic = j.get("hbm_bytes_per_launch")
        secondary = None
        vc = os.path.join(ROOT, "profiles", "valu_count.json")
        if os.path.exists(vc):
            with open(vc) as f:
                j = json.load(f)
            same = j.get("kernel_header_sha256") == sha
            if j.get("kernel") == h.kernel_name and not same:
                secondary = {"bound": "valu_f32", "achieved": None,
                             "source": {"file": "profiles/valu_count.json", "measured_in_this_run": False,
                                        "kernel_header_sha256": j.get("kernel_header_sha256"), "matches_current_source": False}}
            if j.get("kernel") == h.kernel_name and same:
                # the roof that actually binds at SF12 F=2 (SURVEY 8d asks for it beside the HBM one)
                ach = j["valu_wave_instr_per_symbol"] * count / (kernel_ms * 1e-3) / 1e9
                secondary = {"bound": "valu_f32", "achieved": round(ach, 1), "peak": j["peak_spec_g_wave_instr_per_s"],
                             "unit": "G wave-instr/s", "frac": round(ach / j["peak_spec_g_wave_instr_per_s"], 4),
                             "wave_instr_per_symbol": j["valu_wave_instr_per_symbol"],
                             "source": {"file": "profiles/valu_count.json", "measured_in_this_run": False,
                                        "kernel_header_sha256": j.get("kernel_header_sha256"), "matches_current_source": True}}
                if "valu_only_ms_per_launch" in j:   # the same kernel with its loads and LDS exchanges compiled out (LORA_ABL=33):
                    # what a perfectly hidden memory system would leave -- the distance to the compute bound
                    ceil_gbs = j["valu_only_windows_per_launch"] * bytes_per_sym / (j["valu_only_ms_per_launch"] * 1e-3) / 1e9
                    secondary["valu_only_ceiling"] = {"frac_of_hbm_roofline": round(ceil_gbs / HBM_PEAK_GBS, 4),
                                                      "ms_per_launch": j["valu_only_ms_per_launch"],
                                                      "windows_per_launch": j["valu_only_windows_per_launch"],
                                                      "this_run_frac_of_ceiling": round(achieved / ceil_gbs, 4),
                                                      "note": j.get("valu_only_note")}
                if "lds_bytes_per_symbol" in j:      # the other secondary roof SURVEY 8d names: LDS bandwidth
                    lds = j["lds_bytes_per_symbol"] * count / (kernel_ms * 1e-3) / 1e9
                    secondary["lds"] = {"bound": "lds", "achieved": round(lds, 1), "peak": j["lds_peak_gbs"], "unit": "GB/s",
                                        "frac": round(lds / j["lds_peak_gbs"], 4), "bytes_per_symbol": j["lds_bytes_per_symbol"]}
        line = {
            "metric": "demodulated symbols/sec (Msym/s) at SF12 fft_factor=2",
            "value": round(value, 3), "unit": "Msym/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_launches": settle_launches, "kernel_ms_per_step": step_stats,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"SF12 fft_factor=2 beta=25, 10 s x 125 kS/s synthetic chirp capture "
                                   f"(1.25 M samples, 64-symbol packets) x {replicas} replicas per GPU resident in HBM "
                                   f"({n_samples * 8 / 2**30:.2f} GiB), {count} stride-N symbol windows per step per GPU",
                       "kernel": h.kernel_name, "parallelism": f"independent captures per GPU x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "valu_only_ceiling": (secondary or {}).get("valu_only_ceiling", {}).get("frac_of_hbm_roofline"),
                         "kernel_ms": round(kernel_ms, 4), "bytes_per_symbol": bytes_per_sym,
                         "symbols_per_launch": count,
                         "device_copy_gbs": round(copy_gbs, 1), "frac_of_device_copy": round(achieved / copy_gbs, 4),
                         "secondary": secondary},
        }
        if world == 1:
            try:
                line["next_rows"] = {"channeliser": channeliser_leg(torch, g, dev, stream)}
            except Exception as e:              # the headline line must not depend on this leg
                line["next_rows"] = {"channeliser": {"error": str(e)[:200]}}
            try:
                line["next_rows"]["block"] = block_leg(torch, g, dev)
            except Exception as e:
                line["next_rows"]["block"] = {"error": str(e)[:200]}
        if cpu is not None:                              # reported at N=1 only; measured before the GPU was touched
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
