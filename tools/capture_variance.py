#!/usr/bin/env python3
"""Where do slow bench runs come from?  One process captures the temporal frame graph several times (the graph cache is
dropped between attempts) and replays each capture for 8 frames: mean, per-frame GPU time (events) and per-frame HOST time
of the enqueue (no sync).  Finding (profiles/r02_step_jitter.txt): every capture replays at the same speed; slow means come
from bursts of individual slow frames (50-60 ms), i.e. from the box, not from the graph.  (GPU box.)
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
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
        host = []
        for i in range(8):
            ev[i].record()
            h0 = time.perf_counter()
            runner.step()
            host.append(1e3 * (time.perf_counter() - h0))
        ev[8].record()
        torch.cuda.synchronize()
        steps = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(8))
        host.sort()
        res.append((1e3 * (time.perf_counter() - t0) / 8, steps[0], steps[4], steps[-1], host[0], host[4], host[-1]))
        if runner.t > 8:
            runner.t, runner.state = 0, None
for r in res:
    print(f"capture: mean {r[0]:.2f} ms per frame; GPU per-frame min {r[1]:.2f} median {r[2]:.2f} max {r[3]:.2f}; "
          f"host enqueue per frame min {r[4]:.2f} median {r[5]:.2f} max {r[6]:.2f}")
