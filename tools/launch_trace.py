"""Per-launch kernel time of the first launches of a process (developer tool): how long until the steady rate.
   python tools/launch_trace.py [sf f [n_launches [idle_ms]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_lora_amd as g

sf = int(sys.argv[1]) if len(sys.argv) > 1 else 12
f = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 200
idle_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
n = 1 << sf
dev = torch.device("cuda", 0)
torch.manual_seed(1)
iq = torch.view_as_complex(torch.randn((1 << 27), 2, device=dev) * 0.01)
count = iq.numel() // n
oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
h = g.HipDemod(sf, f, 25.0, device=0)
s = torch.cuda.Stream(device=dev); h.set_stream(s.cuda_stream)
torch.cuda.synchronize()
if idle_ms:
    time.sleep(idle_ms * 1e-3)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(nl + 1)]
ev[0].record(s)
for i in range(nl):
    h.batch_strided(iq, 0, n, count, 0, 0, g.capi.WIN_UP, out_idx=oi, out_val=ov, async_=True)
    ev[i + 1].record(s)
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(nl)]
print(f"SF{sf} F{f}: first 10 launches (ms):", " ".join(f"{x:.3f}" for x in ms[:10]))
for a in range(0, nl, 10):
    blk = ms[a:a + 10]
    print(f"  launches {a:4d}-{a + len(blk) - 1:4d}: mean {sum(blk) / len(blk):.4f} ms  min {min(blk):.4f}  max {max(blk):.4f}")
