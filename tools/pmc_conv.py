#!/usr/bin/env python3
"""Runs the gru08.zr-shaped conv (or conv128->128 with arg 'lin') a few times eagerly: a target for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "zr"
H, W = 120, 160
if kind == "zr":
    w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
    pc = ops.pack_conv(w, torch.zeros(256, device=dev), "f16x3")
    xs = [torch.randn(1, 128, H, W, generator=gen).to(dev) for _ in range(3)]
    h = torch.randn(1, 128, H, W, generator=gen).to(dev)
    run = lambda: ops.gru_gates(pc, xs, h)
else:
    w = (torch.randn(128, 128, 3, 3, generator=gen) * 0.02).to(dev)
    pc = ops.pack_conv(w, torch.zeros(128, device=dev), "f16x3")
    xs = [torch.randn(1, 128, H, W, generator=gen).to(dev)]
    run = lambda: ops.conv2d(pc, xs, act="relu")
for _ in range(12):
    run()
torch.cuda.synchronize()
