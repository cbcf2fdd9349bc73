#!/usr/bin/env python3
"""Diagnostic: B independent sequences stacked along the batch dimension must give what each gives alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
import bench
from tcs_mi355 import synth
from tcs_mi355.harness import InputPadder

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W, iters = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (128, 160, 4)
dev = torch.device("cuda:0")
model, _ = bench.build_model(dev)
seqs = [synth.make_sequence(s, n_frames=3, height=H, width=W, max_disp=64 if W < 640 else 192) for s in range(B)]

def frame_inputs(t, idx):
    i1 = torch.stack([torch.as_tensor(seqs[s].frames[t].image1) for s in idx]).to(dev)
    i2 = torch.stack([torch.as_tensor(seqs[s].frames[t].image2) for s in idx]).to(dev)
    K = torch.stack([torch.as_tensor(seqs[s].K) for s in idx]).to(dev)
    T = torch.stack([torch.as_tensor(seqs[s].frames[t].T) for s in idx]).to(dev)
    base = torch.tensor([seqs[s].baseline for s in idx], device=dev)
    padder = InputPadder(i1.shape, divis_by=32)
    (i1, i2), K = padder.pad(i1, i2, K=K)
    return i1.contiguous(), i2.contiguous(), K, T, base

def run(idx):
    outs, state = [], None
    for t in range(3):
        i1, i2, K, T, base = frame_inputs(t, idx)
        params = None
        if state is not None:
            fq, nets, fmap1, pT = state
            params = dict(K=K, T=T, previous_T=pT, last_disp=fq, last_net_list=nets, fmap1=fmap1, baseline=base)
        out = model(i1, i2, iters=iters, test_mode=True, params=params)
        state = (out["flow_q"], out["net_list"], out["fmap1"], T)
        outs.append(out["flow"].clone())
    return outs

batched = run(list(range(B)))
worst = 0.0
for s in range(B):
    single = run([s])
    for t in range(3):
        d = float((batched[t][s] - single[t][0]).abs().max())
        worst = max(worst, d)
        print(f"seq {s} frame {t}: max |batched - single| = {d:.3e}", flush=True)
print("worst", worst)
# throughput
torch.cuda.synchronize(); t0 = time.time(); n = 0
for _ in range(3):
    run(list(range(B))); n += 3 * B
torch.cuda.synchronize(); dt = time.time() - t0
print(f"B={B}: {n / dt:.2f} pairs/s ({dt / (n / B) * 1e3:.2f} ms per batched frame)")
