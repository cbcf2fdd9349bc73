#!/usr/bin/env python3
"""Run-to-run determinism of one non-temporal frame (no splat atomics on that path): the same inputs through the eager
schedule N times and through the HIP-graph replay N times; prints the max abs difference of the up-sampled disparity
against the first eager run.  A schedule without races gives exact zeros in every column (GPU box).
usage: determinism_check.py [iters] [runs] ; env TCS_MI355_X / TCS_MI355_STREAMS select the schedule."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from argparse import Namespace
from core.tc_stereo import TCStereo
from tcs_mi355 import synth, weights

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
args = Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2, context_norm="none",
                 slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
m = TCStereo(args)
weights.load_synth_weights(m)
m = m.to(dev).eval()
seq = synth.make_sequence(2000, n_frames=1, height=480, width=640, max_disp=192.0)
f = seq.frames[0]
i1, i2 = (torch.as_tensor(getattr(f, k)).to(dev).float()[None] for k in ("image1", "image2"))
outs, sums = [], []
with torch.no_grad():
    for use_graph in (False, True):
        m.use_hip_graph = use_graph
        for r in range(runs):
            if not use_graph:
                m._checksums = []
            outs.append((("graph" if use_graph else "eager"), m(i1, i2, iters=iters, test_mode=True)["flow"].clone()))
            if not use_graph:
                sums.append(m._checksums)
                m._checksums = None
torch.cuda.synchronize()
# first intermediate tensor (iteration order) whose device-side sum differs between eager run 0 and a later eager run
for r in range(1, len(sums)):
    for it, (a, b) in enumerate(zip(sums[0], sums[r])):
        bad = [k for k in a if float(a[k]) != float(b[k])]
        if bad:
            print(f"eager run {r}: first divergence in iteration {it}: {bad}")
            break
ref = outs[0][1]
print(f"X=[{os.environ.get('TCS_MI355_X', '')}] STREAMS={os.environ.get('TCS_MI355_STREAMS', '1')} iters={iters}: " +
      " ".join(f"{k}:{float((o - ref).abs().max()):.1e}" for k, o in outs))
