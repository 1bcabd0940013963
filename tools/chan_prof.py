"""A few launches of the channeliser on 1 GiB resident in HBM: target for rocprofv3 (developer tool).
  python tools/chan_prof.py [rate] [offset] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gr_lora_amd as g
rate = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
offset = float(sys.argv[2]) if len(sys.argv) > 2 else -0.25
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = (1 << 30) // 8
x = torch.view_as_complex(torch.randn(n, 2, device="cuda:0", dtype=torch.float32))
ch = g.channelizer(2 * np.pi * offset, rate)
out = torch.empty(ch.output_count(n) + 8, dtype=torch.complex64, device="cuda:0")
for _ in range(reps):
    y = ch.work(x, out=out)
torch.cuda.synchronize()
print("done", ch.kernel_name, n, len(y))
