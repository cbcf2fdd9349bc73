#!/usr/bin/env python3
"""Soak: N frames of graph replay at 640x480 / 32 iterations; every output must stay finite and non-positive (flow = -disp),
and frames 10..N must match frames 0..9 of the first pass through the clip (the clip loops, the state is reset at each wrap)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
import bench
from tcs_mi355 import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev)
seq = synth.make_sequence(2000, n_frames=10, height=480, width=640, max_disp=192.0)
r = bench.ClipRunner(model, seq, dev, 32)
first = []
worst = 0.0
per_pos = [0.0] * 10
t0 = time.time()
with torch.no_grad():
    for i in range(n):
        flow = r.step()["flow"]
        assert torch.isfinite(flow).all() and float(flow.max()) <= 0.0, i
        if i < 10:
            first.append(flow.clone())
        else:
            d = float((flow - first[i % 10]).abs().mean())
            worst = max(worst, d)
            per_pos[i % 10] = max(per_pos[i % 10], d)
torch.cuda.synchronize()
print("max mean |difference| by position in the clip:", " ".join(f"{v:.1e}" for v in per_pos))
print(f"{n} frames, {n / (time.time() - t0):.2f} pairs/s incl. checks, worst mean |difference| to the first pass {worst:.3e}")
