"""Launch the strided hot path a few times for one (sf, F): target for rocprofv3 (developer tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gr_lora_amd as g
sf, f = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
total = int(sys.argv[4]) if len(sys.argv) > 4 else (1 << 30)
n = 1 << sf
count = total // (8 * n)
dev = torch.device("cuda:0")
iq = torch.randn(count * n, 2, device=dev, dtype=torch.float32)
oi = torch.empty(count, dtype=torch.int16, device=dev); ov = torch.empty(count, dtype=torch.float32, device=dev)
h = g.HipDemod(sf, f)
for _ in range(reps):
    h.batch_strided(iq, 0, n, count, out_idx=oi, out_val=ov)
torch.cuda.synchronize()
print("done", sf, f, count)
