#!/usr/bin/env python3
"""How many conv blocks does a CU run concurrently?  Times the gru08.zr-like kernel at ~1, ~2, ~3 blocks per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
for math in ("f16x3", "f32"):
    pc = ops.pack_conv(w, torch.zeros(256, device=dev), math)
    for rows4 in (3, 6, 13, 26, 39, 52, 78):
        H, W = 4 * rows4, 160
        xs = [torch.randn(1, 128, H, W, generator=gen).to(dev) for _ in range(3)]
        h = torch.randn(1, 128, H, W, generator=gen).to(dev)
        for _ in range(3):
            ops.gru_gates(pc, xs, h)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.gru_gates(pc, xs, h)
        e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) * 100
        print(f"{math}: H={H:4d} patches={rows4*5:4d} (x cout tiles) -> {us:8.1f} us", flush=True)
