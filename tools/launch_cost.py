"""What one small batch call costs (developer tool): the host driver pays this per cache miss."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gr_lora_amd as g

sf, f, n = 10, 2, 1024
dev = torch.device("cuda:0")
h = g.HipDemod(sf, f)
x = (np.random.default_rng(1).standard_normal(400 * n) + 0j).astype(np.complex64)
h.stage_iq(x)
t_iq = torch.from_numpy(x).to(dev)
for count in (16, 176, 400, 4096):
    reps = 300
    cnt = min(count, 399)
    oi = np.empty(cnt, np.uint16); ov = np.empty(cnt, np.float32)
    for _ in range(20):
        h.batch_strided(None, 0, n, cnt, out_idx=oi, out_val=ov)
    t0 = time.perf_counter()
    for _ in range(reps):
        h.batch_strided(None, 0, n, cnt, out_idx=oi, out_val=ov)
    t_host = (time.perf_counter() - t0) / reps
    di = torch.empty(cnt, dtype=torch.int16, device=dev); dv = torch.empty(cnt, dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        h.batch_strided(t_iq, 0, n, cnt, out_idx=di, out_val=dv)            # synchronous, results stay on the device
    t_dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        h.batch_strided(t_iq, 0, n, cnt, out_idx=di, out_val=dv, async_=True)
    torch.cuda.synchronize()
    t_async = (time.perf_counter() - t0) / reps
    print(f"{cnt:5d} windows: host results {t_host*1e6:6.1f} us | device results, synchronous {t_dev*1e6:6.1f} us | enqueue only {t_async*1e6:6.1f} us", flush=True)
h.close()
