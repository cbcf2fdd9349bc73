#!/usr/bin/env python3
"""us per launch of the fused HiddenstateUpdater kernel at the loop's size (120x160), graph-replayed burst (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import s16

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=gen).to(dev)
B = int(os.environ.get("BATCH", "1"))
H, W = 120, 160
h = s16.to_s16(torch.tanh(R(B, 128, H, W)))
delta = R(B, 1, H, W)
w = (R(64), R(64) * 0.1, s16.pack_frags(R(64, 64, 1, 1) * 0.15, R(64) * 0.1, 64), s16.pack_frags(R(256, 192, 1, 1) * 0.08, R(256) * 0.1, 0),
     s16.pack_frags(R(128, 192, 1, 1) * 0.08, R(128) * 0.1, 0))
run = lambda: s16.hidden_update(h, delta, *w)
for _ in range(3):
    run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for _ in range(100):
            run()
g.replay(); torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    g.replay()
e.record(); torch.cuda.synchronize()
print(f"k_hidden_update_s16 batch {B}: {a.elapsed_time(e) * 1e3 / 300:.1f} us per launch")
