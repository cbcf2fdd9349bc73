#!/usr/bin/env python3
"""Diagnostic: whole-frame capture with TCS_MI355_STREAMS=1 (forked branches)."""
import os, sys, faulthandler, time
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TCS_MI355_STREAMS"] = "1"
import tcs_paths; tcs_paths.add_product_path()
import torch
import bench
from tcs_mi355 import synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 160)
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev)
seq = synth.make_sequence(0, n_frames=3, height=H, width=W, max_disp=64)
runner = bench.ClipRunner(model, seq, dev, iters)
for i in range(6):
    t = time.time(); runner.step(); torch.cuda.synchronize(); print("frame", i, "ms", (time.time() - t) * 1e3, flush=True)
