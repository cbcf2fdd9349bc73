#!/usr/bin/env python3
"""(Uses the CPU oracle, hence kept under tests/.)  Free-running (not teacher-forced) 2-frame comparison: eager HIP, graph HIP, oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tcs_paths; tcs_paths.add_product_path()
import torch
import tcs_oracle as oracle
from bench import build_model
from tcs_mi355 import synth
from tcs_mi355.harness import run_sequence
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model, W = build_model(dev)
torch.set_num_threads(16)
seq = synth.make_sequence(2000, n_frames=3, height=480, width=640, max_disp=192.0)
res = {}
for mode in ("eager", "graph", "graph2"):
    model.use_hip_graph = mode != "eager"
    p = []
    run_sequence(model, seq, iters=iters, device=dev, collect=p)
    res[mode] = [t.cpu() for t in p]
ref = []
run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(W, a, b, iters=kw["iters"], params=kw["params"]), seq, iters=iters,
             device=torch.device("cpu"), collect=ref)
for t in range(3):
    print(f"frame {t}: |ref| {ref[t].abs().mean():.4f}  " + "  ".join(
        f"{m}-oracle {float((res[m][t]-ref[t]).abs().mean()):.3e}" for m in res) +
        f"  eager-graph {float((res['eager'][t]-res['graph'][t]).abs().mean()):.3e}  graph-graph2 {float((res['graph'][t]-res['graph2'][t]).abs().mean()):.3e}", flush=True)
