#!/usr/bin/env python3
"""Is the frame time a property of the process or of the capture?  One process captures the temporal frame graph several
times (the graph cache is dropped between attempts) and times replays of each capture.  (GPU box.)
usage: capture_variance.py [captures] ; env TCS_MI355_X / TCS_MI355_FORK_SITES select the schedule."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
from tcs_mi355 import native, synth
native.lib()
model, _ = bench.build_model(dev)
model.use_hip_graph = True
seq = synth.make_sequence(2000, n_frames=bench.CLIP_LEN, height=bench.HEIGHT, width=bench.WIDTH, max_disp=bench.MAX_DISP)
runner = bench.ClipRunner(model, [seq], dev, bench.ITERS)
res = []
with torch.no_grad():
    for k in range(n):
        model._graphs = None                        # drop the cache: the next two frames capture both branches again
        runner.t, runner.state = 0, None
        runner.step(); runner.step(); runner.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            runner.step()
        torch.cuda.synchronize()
        res.append(1e3 * (time.perf_counter() - t0) / 8)
        if runner.t > 8:
            runner.t, runner.state = 0, None
print("ms per frame for each capture:", " ".join(f"{r:.2f}" for r in res))
