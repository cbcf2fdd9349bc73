#!/usr/bin/env python3
"""Micro-benchmark of tcs_conv2d on the refinement loop's layer shapes (GPU box).  A burst of launches is captured
into a HIP graph and the replay is timed with HIP events (eager launches cost 10-14 us each from Python, more than the
small layers themselves); prints us per launch and effective TFLOP/s (2*MACs, fp32-equivalent)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops

dev = torch.device("cuda:0")
SHAPES = [  # name, (cins...), cout, k, H, W, epilogue
    ("gru08.zr", (128, 128, 128), 256, 3, 120, 160, "zr"),
    ("gru08.q", (128, 128, 128), 128, 3, 120, 160, "q"),
    ("gru16.zr", (128, 128, 128), 256, 3, 60, 80, "zr"),
    ("gru32.zr", (128, 128), 256, 3, 30, 40, "zr"),
    ("conv128->128", (128,), 128, 3, 120, 160, "lin"),
    ("conv192->128", (96, 96), 128, 3, 120, 160, "lin"),
    ("conv128->256", (128,), 256, 3, 120, 160, "lin"),
    ("conv64->64", (64,), 64, 3, 120, 160, "lin"),
    ("conv256->1", (256,), 1, 3, 120, 160, "lin"),
    ("1x1 192->256", (128, 64), 256, 1, 120, 160, "zr"),
    ("1x1 27->96", (27,), 96, 1, 120, 160, "lin"),
    ("conv128->128/16", (128,), 128, 3, 60, 80, "lin"),
    ("conv128->128/32", (128,), 128, 3, 30, 40, "lin"),
    ("conv96->96/16", (96,), 96, 3, 60, 80, "lin"),
    ("conv256->128/8", (128, 128), 128, 3, 120, 160, "lin"),
]
maths = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f16x3", "f32"]
gen = torch.Generator().manual_seed(0)
ONLY = [t for t in os.environ.get("BENCH_ONLY", "").split(",") if t]
for name, cins, cout, k, H, W, epi in SHAPES:
    if ONLY and name not in ONLY:
        continue
    cin = sum(cins)
    w = (torch.randn(cout, cin, k, k, generator=gen) * 0.02).to(dev)
    b = torch.zeros(cout, device=dev)
    xs = [torch.randn(1, c, H, W, generator=gen).to(dev) for c in cins]
    hid = cout // 2 if epi == "zr" else cout
    h = torch.randn(1, hid, H, W, generator=gen).to(dev)
    z = torch.rand(1, hid, H, W, generator=gen).to(dev)
    line = f"{name:14s} {H}x{W} cin {cin:4d} cout {cout:4d}: "
    for m in maths:
        pc = ops.pack_conv(w, b, m)
        def run():
            if epi == "zr":
                ops.gru_gates(pc, xs, h)
            elif epi == "q":
                ops.gru_update(pc, xs, h, z)
            else:
                ops.conv2d(pc, xs, act="relu")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        n = 200
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    run()
        g.replay()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        a.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) * 1e3 / (n * reps)
        tf = 2.0 * H * W * cin * cout * k * k / us / 1e6
        line += f"{m}: {us:8.1f} us {tf:7.1f} TF/s   "
    print(line, flush=True)
