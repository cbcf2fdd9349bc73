#!/usr/bin/env python3
"""Runs one convolution shape on the S16 kernel (tcs_conv2d_s16, the tile the library chooses) a few times eagerly: a target for
rocprofv3 --pmc passes.  usage: pmc_conv_s16.py [zr | lin128]
  zr     (default) gru08.zr: 384 -> 256, 3x3, GRU_ZR epilogue at 120x160 (profiles/r02_conv_gru08zr_pmc.txt)
  lin128 128 -> 128, 3x3, LINEAR epilogue (ReLU, S16 output) at 120x160 — the commonest 1/4-scale layer of the loop
         (profiles/r03_conv128_occupancy_pmc.txt: the same launch on two builds of the library, TCS_MI355_LIB=; CFG= fixes the tile)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops, s16
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
H, W = 120, 160
mode = sys.argv[1] if len(sys.argv) > 1 else "zr"
if mode == "lin128":
    w = (torch.randn(128, 128, 3, 3, generator=gen) * 0.02).to(dev)
    pc = ops.pack_conv(w, torch.zeros(128, device=dev), "f16x3")
    x = s16.to_s16(torch.randn(1, 128, H, W, generator=gen).to(dev))
    out = s16.zeros(1, 128, H, W, dev)
    for _ in range(12):
        s16.conv2d(pc, [x], act="relu", out16=out, tile_cfg=int(os.environ.get("CFG", "0")))
else:
    w = (torch.randn(256, 384, 3, 3, generator=gen) * 0.02).to(dev)
    pc = ops.pack_conv(w, torch.zeros(256, device=dev), "f16x3")
    xs = [s16.to_s16(torch.randn(1, 128, H, W, generator=gen).to(dev)) for _ in range(3)]
    h = s16.to_s16(torch.randn(1, 128, H, W, generator=gen).to(dev))
    z, rh = torch.empty(1, 128, H, W, device=dev), s16.zeros(1, 128, H, W, dev)
    for _ in range(12):
        s16.gru_gates(pc, xs, h, z_out=z, rh_out=rh)
torch.cuda.synchronize()
