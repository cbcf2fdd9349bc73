#!/usr/bin/env python3
"""Runs the gru08.zr-shaped convolution on the S16 kernel (tcs_conv2d_s16, the loop's configuration chosen by the library) a few
times eagerly: a target for rocprofv3 --pmc passes (profiles/r02_conv_gru08zr_pmc.txt)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops, s16
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
H, W = 120, 160
w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
pc = ops.pack_conv(w, torch.zeros(256, device=dev), "f16x3")
xs = [s16.to_s16(torch.randn(1, 128, H, W, generator=gen).to(dev)) for _ in range(3)]
h = s16.to_s16(torch.randn(1, 128, H, W, generator=gen).to(dev))
z, rh = torch.empty(1, 128, H, W, device=dev), s16.zeros(1, 128, H, W, dev)
for _ in range(12):
    s16.gru_gates(pc, xs, h, z_out=z, rh_out=rh)
torch.cuda.synchronize()
