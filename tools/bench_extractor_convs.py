#!/usr/bin/env python3
"""Extractor layer shapes (B=2: left and right image through the shared trunk): our fp16-split conv kernel against
torch.nn.functional.conv2d (MIOpen fp32) on the same tensors, graph-timed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch, torch.nn.functional as F
from tcs_mi355 import ops
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); g.replay(); e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / (2 * n)

for name, cin, cout, H, W, stride in (("layer1 64->64", 64, 64, 480, 640, 1), ("layer2.0 64->96 s2", 64, 96, 480, 640, 2),
                                      ("layer2 96->96", 96, 96, 240, 320, 1), ("layer3.0 96->128 s2", 96, 128, 240, 320, 2),
                                      ("layer3 128->128", 128, 128, 120, 160, 1), ("head 128->256", 128, 256, 120, 160, 1)):
    x = torch.randn(2, cin, H, W, generator=gen).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.05).to(dev)
    b = torch.zeros(cout, device=dev)
    pc = ops.pack_conv(w, b, "f16x3")
    ours = timed(lambda: ops.conv2d(pc, [x], act="relu", stride=stride))
    ref = timed(lambda: torch.relu_(F.conv2d(x, w, b, stride=stride, padding=1)))
    d = float((ops.conv2d(pc, [x], act="relu", stride=stride) - torch.relu(F.conv2d(x, w, b, stride=stride, padding=1))).abs().max())
    print(f"{name:22s} {H}x{W} B=2: ours {ours:7.1f} us   MIOpen conv+relu {ref:7.1f} us   max diff {d:.2e}", flush=True)
