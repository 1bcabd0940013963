"""SFD sweep rate (stride N/16, stepping chirp offset, taps reloaded per window) for a few SF/F: developer tool, used with
LORA_HIP_LIB=build_ab/<ablation>.so to see what the sweep's loads cost (a shared-tile kernel could only save those).
  python tools/sfd_bench.py [sf,f ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2e_bench import sfd_sweep
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(8, 2), (9, 2), (10, 2), (12, 2)]
for sf, f in cfgs:
    sfd_sweep(sf, f)
