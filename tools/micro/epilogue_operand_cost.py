#!/usr/bin/env python3
"""How much of a small convolution launch is the wait for its epilogue operands?  A chain of dependent tcs_conv2d_s16 launches
(x -> y -> x -> ..., 64 -> 64 channels, 3x3, S16 in / S16 out) replayed as a HIP graph, timed per launch for: no bias / bias /
bias + fp32 addend, on a 1/8-scale grid (135 workgroups: one round, latency-bound) and a 1/4-scale grid (600).
The difference between the rows is what requesting those operands BEFORE the K loop could save per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops, s16

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)


def timed(run, n=200, reps=5):
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                run()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3 / n)
    return best


for H, W in ((60, 80), (120, 160)):
    w = (torch.randn(64, 64, 3, 3, generator=gen) * 0.02).to(dev)
    b = (torch.randn(64, generator=gen) * 0.1).to(dev)
    x, y = s16.to_s16(torch.randn(1, 64, H, W, generator=gen).to(dev)), s16.zeros(1, 64, H, W, dev)
    add = torch.randn(1, 64, H, W, generator=gen).to(dev)
    bufs = [x, y]
    for name, pc, addend in (("no bias", ops.pack_conv(w, None, "f16x3"), None), ("bias", ops.pack_conv(w, b, "f16x3"), None),
                             ("bias + fp32 addend", ops.pack_conv(w, b, "f16x3"), add)):
        state = {"i": 0}

        def run():
            i = state["i"]; state["i"] = i ^ 1
            s16.conv2d(pc, [bufs[i]], act="relu", addend=addend, post_scale=0.5, out16=bufs[i ^ 1])
        print(f"{H}x{W} 64->64 3x3  {name:20s} {timed(run):6.2f} us per launch (chained, graph replay)", flush=True)
