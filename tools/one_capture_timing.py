import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, gr_lora_amd as g, oracle
sf, f, n = 10, 2, 1024
caps = [np.concatenate([oracle.modulate(sf, oracle.random_symbols(sf, 256, 4000 + c), 0x12), np.zeros(4 * n, np.complex64)]) for c in range(64)]
blk = g.demod(sf, False, 25.0, f)
for rep in range(3):
    t0 = time.perf_counter(); L = 0
    for c in caps:
        blk.work(c); L += blk.stats["launches"]; blk.reset()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt/len(caps)*1e6:.0f} us per capture, {L/len(caps):.1f} launches per capture, {caps[0].nbytes/1e6:.2f} MB", flush=True)
import torch
x = torch.from_numpy(caps[0]); d = torch.empty_like(x, device="cuda")
for pin in (False, True):
    src = x.pin_memory() if pin else x
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): d.copy_(src, non_blocking=True); torch.cuda.synchronize()
    print("H2D", "pinned" if pin else "pageable", (time.perf_counter() - t0) / 50 * 1e6, "us per 2.3 MB copy")
