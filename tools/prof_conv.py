#!/usr/bin/env python3
"""Runs one conv shape repeatedly (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops
dev = torch.device("cuda:0")
math = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
gen = torch.Generator().manual_seed(0)
w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
xs = [torch.randn(1, 128, 120, 160, generator=gen).to(dev) for _ in range(3)]
h = torch.randn(1, 128, 120, 160, generator=gen).to(dev)
pc = ops.pack_conv(w, torch.zeros(256, device=dev), math)
for _ in range(10):
    ops.gru_gates(pc, xs, h)
torch.cuda.synchronize()
