#!/usr/bin/env python3
"""How many conv workgroups does a CU run concurrently?  Times the gru08.zr kernel (graph replay) as the grid grows:
the time steps up each time the grid exceeds 256 x (resident workgroups per CU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
pc = ops.pack_conv(w, torch.zeros(256, device=dev), "f16x3")
ROWS4 = tuple(int(v) for v in os.environ.get("PROBE_ROWS4", "3,6,9,12,13,16,19,22,25,26,28,30,32,36,38,39,42,48,51,52,56,64,77,78,90,102,103,116,128").split(","))
for rows4 in ROWS4:
    H, W = 4 * rows4, 160
    xs = [torch.randn(1, 128, H, W, generator=gen).to(dev) for _ in range(3)]
    h = torch.randn(1, 128, H, W, generator=gen).to(dev)
    for _ in range(2):
        ops.gru_gates(pc, xs, h)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    n = 50
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                ops.gru_gates(pc, xs, h)
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) * 1e3 / (2 * n)
    mt = int(os.environ.get("TCS_F16_MT", "0")) or (2 if rows4 * 5 * 4 >= 512 else 1)
    blocks = rows4 * 5 * (8 // mt)
    print(f"H={H:4d} workgroups={blocks:5d} ({blocks / 256:5.2f} per CU, MT={mt}) -> {us:8.1f} us  ({us / blocks * 256:6.1f} us per CU-round)", flush=True)
