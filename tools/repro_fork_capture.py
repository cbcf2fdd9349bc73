#!/usr/bin/env python3
"""Diagnostic: does a forked (multi-stream) capture survive hipStreamEndCapture?  usage: repro_fork_capture.py KIND N"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
os.environ["TCS_MI355_STREAMS"] = "1"
import torch
from tcs_mi355 import ops, streams

kind, n = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
x = torch.randn(1, 64, 120, 160, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
pc = ops.pack_conv(w, torch.zeros(64, device=dev), "f16x3")

def body():
    y = x
    for _ in range(n):
        if kind == "conv":
            a, b = streams.fork_join([lambda: ops.conv2d(pc, [y], act="relu"), lambda: ops.conv2d(pc, [y], act="relu")])
        elif kind == "torch":
            a, b = streams.fork_join([lambda: torch.relu(y) * 0.5, lambda: torch.tanh(y)])
        elif kind == "nested":
            def inner():
                p, q = streams.fork_join([lambda: ops.conv2d(pc, [y], act="relu"), lambda: ops.conv2d(pc, [y], act="relu")])
                return p + q
            a, b = streams.fork_join([inner, lambda: ops.conv2d(pc, [y], act="relu")])
        y = a * 0.5 + b * 0.5
    return y

side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ref = body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body()
print("captured", kind, n, flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed; max diff vs eager", float((out - ref).abs().max()), flush=True)
