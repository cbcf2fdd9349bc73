#!/usr/bin/env python3
"""What separates the lookup's 4.7 us inside the frame from the 1.84 us of a hot burst?  (GPU box.)
Every lookup of the frame is launched TWICE back to back (the second into a scratch tensor, same arguments otherwise) with the device-clock
stamps on; the first launch of a pair meets what the frame leaves it (cold code, fresh coordinates, whatever is left of the pyramid), the
second meets exactly what the first touched.  Prints the mean in-kernel span of the first and of the second launches.
usage: lookup_double.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

dev = torch.device("cuda:0")
from tcs_mi355 import native, ops, synth
import core.corr as corr_mod
native.lib()
model, _ = bench.build_model(dev)
model.use_hip_graph = True
seq = synth.make_sequence(2000, n_frames=bench.CLIP_LEN, height=bench.HEIGHT, width=bench.WIDTH, max_disp=bench.MAX_DISP)

orig = corr_mod.CorrBlock1D.__call__
scratch = {}
def twice(self, coords):
    out = orig(self, coords)
    key = tuple(out.shape)
    if key not in scratch:
        scratch[key] = torch.empty_like(out)
    ops.corr_lookup(self._pyr, coords[:, :1].float().contiguous(), self.radius, out=scratch[key])
    return out
corr_mod.CorrBlock1D.__call__ = twice

probe = ops.LookupProbe(dev, slots=64)
ops.LOOKUP_PROBE = probe
runner = bench.ClipRunner(model, [seq], dev, bench.ITERS, prefetch=True)
first, second = [], []
with torch.no_grad():
    for _ in range(3):
        runner.step()
    probe.reset()
    for _ in range(bench.CLIP_LEN):
        runner.step()
        snap = probe.buf.clone()
        probe.reset()
        torch.cuda.synchronize()
        buf = snap.cpu()
        start, end = buf[..., 0], buf[..., 1]
        big = torch.iinfo(torch.int64).max
        s = torch.where(start > 0, start, torch.full_like(start, big)).min(dim=1).values
        e = end.max(dim=1).values
        d = ((e - s).double() * 0.01).tolist()
        used = (end > 0).any(dim=1).tolist()
        for k, (dk, u) in enumerate(zip(d, used)):
            if u:
                (first if k % 2 == 0 else second).append(dk)      # 64 slots, 2 launches per iteration: even slot = first of the pair
print(f"first launch of a pair  (as the frame leaves things): mean {np.mean(first):.2f} us  median {np.median(first):.2f}  n={len(first)}")
print(f"second launch of a pair (right behind the first)    : mean {np.mean(second):.2f} us  median {np.median(second):.2f}  n={len(second)}")
