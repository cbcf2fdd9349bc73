#!/usr/bin/env python3
"""Host-side clock of one frame's calls (GPU box): how long each enqueue call of FrameGraphs.__call__ / prefetch takes on the host and how
far the host runs ahead of the GPU.  Wraps the graph replays, the state copies and the output clones with perf_counter stamps (no sync inside
the measured region) and prints medians over `n` steady-state frames, plus the host's lead at the start of each frame (= how much enqueued GPU
work is still pending when forward() is called: measured as the time a synchronize() at that point would take, on a SEPARATE pass).
usage: host_phases.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
from tcs_mi355 import native, synth
native.lib()
model, _ = bench.build_model(dev)
model.use_hip_graph = True
seq = synth.make_sequence(2000, n_frames=bench.CLIP_LEN, height=bench.HEIGHT, width=bench.WIDTH, max_disp=bench.MAX_DISP)
runner = bench.ClipRunner(model, [seq], dev, bench.ITERS)
with torch.no_grad():
    for _ in range(4):
        runner.step()
    torch.cuda.synchronize()
    # patch CUDAGraph.replay with a timing wrapper
    import torch.cuda.graphs as G
    stamps = []
    orig = torch.cuda.CUDAGraph.replay
    def timed(self):
        t0 = time.perf_counter()
        orig(self)
        stamps.append(("replay", 1e3 * (time.perf_counter() - t0)))
    torch.cuda.CUDAGraph.replay = timed
    per_frame = []
    t_prev = time.perf_counter()
    for i in range(n):
        stamps.clear()
        h0 = time.perf_counter()
        runner.step()
        h1 = time.perf_counter()
        per_frame.append((1e3 * (h1 - h0), [s[1] for s in stamps]))
    torch.cuda.synchronize()
    torch.cuda.CUDAGraph.replay = orig
    host = np.array([p[0] for p in per_frame])
    reps = np.array([p[1] for p in per_frame if len(p[1]) == 3])
    print(f"host time per step(): median {np.median(host):.2f} ms (min {host.min():.2f}, max {host.max():.2f})")
    if len(reps):
        print("graph replays per step [head, loop, extract(prefetch)] median ms:", np.round(np.median(reps, 0), 3).tolist())
    # lead: how long does a sync at the START of a step take (pending GPU work)
    leads = []
    for i in range(10):
        t0 = time.perf_counter(); torch.cuda.synchronize(); leads.append(1e3 * (time.perf_counter() - t0))
        runner.step()
    print("pending GPU work when step() is called (sync time, ms):", np.round(leads, 2).tolist(), "(first = after a sync: 0)")
    # and with the sync in the middle: after forward(), before prefetch -> measured by stepping manually

# ---- second pass: is the hole between a frame's output clones and the next frame's EXTRACT a host or a GPU matter? ----------------------------
# events around the prefetch call (GPU clock) + host stamps of the same points; a sync only at the very end
with torch.no_grad():
    torch.cuda.synchronize()
    recs = []
    base_ev = torch.cuda.Event(enable_timing=True); base_ev.record(); base_t = time.perf_counter()
    for i in range(12):
        i1, i2, K, T = runner.frames[runner.t]
        params = None
        if runner.t > 0 and runner.state is not None:
            flow_q, nets, fmap1, prev_T = runner.state
            params = dict(K=K, T=T, previous_T=prev_T, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=runner.baseline)
        h_a = time.perf_counter()
        out = model(i1, i2, iters=bench.ITERS, test_mode=True, params=params)
        h_b = time.perf_counter()
        e_b = torch.cuda.Event(enable_timing=True); e_b.record()          # after forward(t)'s last enqueue (the output clones)
        nxt = (runner.t + 1) % runner.n
        model.prefetch(runner.frames[nxt][0], runner.frames[nxt][1], first=(nxt == 0), inputs_ready=True)
        h_c = time.perf_counter()
        e_c = torch.cuda.Event(enable_timing=True); e_c.record()          # after the prefetch's enqueue (EXTRACT of t+1)
        recs.append((h_a, h_b, h_c, e_b, e_c))
        runner.state = (out["flow_q"], out["net_list"], out["fmap1"], T)
        runner.t = (runner.t + 1) % runner.n
        if runner.t == 0:
            runner.state = None
    torch.cuda.synchronize()
    print("frame: host forward() ms | host prefetch() ms | GPU time from end-of-clones event to end-of-EXTRACT event ms | host lead at the clones event ms")
    for h_a, h_b, h_c, e_b, e_c in recs[2:]:
        g_b = base_ev.elapsed_time(e_b)           # GPU time at which the clones finished, since base
        lead = g_b - 1e3 * (h_b - base_t)         # how much later the GPU reached that point than the host enqueued it
        print(f"  {1e3 * (h_b - h_a):7.2f} | {1e3 * (h_c - h_b):6.2f} | {e_b.elapsed_time(e_c):6.2f} | {lead:7.2f}")
