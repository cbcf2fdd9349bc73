#!/usr/bin/env python3
"""bench.py — stereo-pairs/sec of the TC-Stereo hot path on MI355X.

Metric (BASELINE.json): stereo-pairs/sec at 640x480, D=192, 32 GRU iterations.
A "step" is one stereo pair of a synthetic 10-frame 640x480 sequence through `TCStereo.forward`
(frame 0 takes the argmax branch, frames 1-9 the temporal-warp branch; state resets when the clip
wraps).  Inputs are resident in HBM before the timed region.  One process per GPU, one independent
sequence per rank (no data-path collective; one all_gather of EPE statistics at the end).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 with `roofline` (corr lookup, HIP events in the timed region) and
`cpu_baseline` (the CPU oracle on the host cores, rank 0 / N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tcs_paths  # noqa: E402

tcs_paths.add_product_path()

import numpy as np  # noqa: E402
import torch  # noqa: E402

HEIGHT, WIDTH, MAX_DISP, ITERS, CLIP_LEN = 480, 640, 192.0, 32, 10
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOOKUP_BYTES_PER_PIXEL = 308    # 4 levels x 10 taps x 4 B + 4 B coord + 36 x 4 B out


def build_model(dev):
    from argparse import Namespace

    from core.tc_stereo import TCStereo
    from tcs_mi355.weights import synth_state_dict
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        W = synth_state_dict(json.load(f)["shared_backbone"])
    args = Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2,
                     context_norm="none", slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
    model = TCStereo(args)
    model.load_state_dict(W, strict=True)
    return model.to(dev).eval(), W


class ClipRunner:
    """Steps through a device-resident clip, carrying the temporal state like evaluate_stereo.py:170-197."""

    def __init__(self, model, seq, dev, iters):
        from tcs_mi355.harness import InputPadder
        self.model, self.iters, self.n = model, iters, len(seq.frames)
        K_raw = torch.as_tensor(seq.K, device=dev)[None]
        self.baseline = torch.tensor([seq.baseline], device=dev)
        self.frames = []
        for fr in seq.frames:
            i1, i2 = torch.as_tensor(fr.image1, device=dev)[None], torch.as_tensor(fr.image2, device=dev)[None]
            padder = InputPadder(i1.shape, divis_by=32)
            (i1, i2), K = padder.pad(i1, i2, K=K_raw)
            self.frames.append((i1.contiguous(), i2.contiguous(), K, torch.as_tensor(fr.T, device=dev)[None]))
        self.t = 0
        self.state = None
        self.last = None

    def step(self):
        i1, i2, K, T = self.frames[self.t]
        params = None
        if self.t > 0 and self.state is not None:
            flow_q, nets, fmap1, prev_T = self.state
            params = dict(K=K, T=T, previous_T=prev_T, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=self.baseline)
        out = self.model(i1, i2, iters=self.iters, test_mode=True, params=params)
        self.state = (out["flow_q"], out["net_list"], out["fmap1"], T)
        self.last = out
        self.t = (self.t + 1) % self.n
        if self.t == 0:
            self.state = None
        return out


class LookupTimer:
    """HIP events around every corr-lookup launch of the timed region (same stream as the kernels:
    the library launches on torch's current stream)."""

    def __init__(self):
        from tcs_mi355 import ops
        self.ops, self.orig, self.pairs, self.pixels, self.on = ops, ops.corr_lookup, [], 0, False

    def __enter__(self):
        def timed(pyr, coords, radius=4, out=None):
            if not self.on:
                return self.orig(pyr, coords, radius, out)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = self.orig(pyr, coords, radius, out)
            b.record()
            self.pairs.append((a, b))
            self.pixels = pyr.B * pyr.H * pyr.W
            return r
        self.ops.corr_lookup = timed
        return self

    def __exit__(self, *exc):
        self.ops.corr_lookup = self.orig

    def empty_bracket_us(self, n=200):
        """Cost of an event pair with nothing between (subtracted as calibration)."""
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record()
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev])) * 1e3

    def result(self):
        if not self.pairs:
            return None
        torch.cuda.synchronize()
        us = np.array([a.elapsed_time(b) for a, b in self.pairs]) * 1e3
        cal = self.empty_bracket_us()
        dur_us = max(float(np.mean(us)) - cal, 1e-3)
        alg_bytes = LOOKUP_BYTES_PER_PIXEL * self.pixels
        achieved = alg_bytes / (dur_us * 1e-6) / 1e9
        return {"bound": "hbm", "kernel": "k_corr_lookup<4>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_us": round(dur_us, 3),
                "event_pair_overhead_us": round(cal, 3), "launches": len(self.pairs), "algorithmic_bytes_per_launch": alg_bytes}


def cpu_baseline(W, seq, gpu_preds, n_frames=2):
    """The CPU oracle ("port" of the reference's fp32 CPU path) on the first frames of the same clip."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tcs_oracle as oracle
    from tcs_mi355.harness import run_sequence
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    sub = type(seq)(seq.frames[:n_frames], seq.K, seq.baseline)
    preds = []
    t0 = time.perf_counter()
    run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(W, a, b, iters=kw["iters"], params=kw["params"]), sub, iters=ITERS,
                 device=torch.device("cpu"), collect=preds)
    dt = time.perf_counter() - t0
    epes = [float((g.cpu() - p).abs().mean()) for g, p in zip(gpu_preds, preds)]
    return {"value": round(n_frames / dt, 4), "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
            "sample": f"first {n_frames} frames of the same 640x480 clip, 32 iters, oracle/tcs_oracle.py on torch CPU fp32, "
                      f"{torch.get_num_threads()} threads"}, epes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    from tcs_mi355 import dist as tdist
    from tcs_mi355 import native, synth
    rank, world, local = tdist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    native.lib()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model, W = build_model(dev)
    seq = synth.make_sequence(2000 + rank, n_frames=CLIP_LEN, height=HEIGHT, width=WIDTH, max_disp=MAX_DISP)
    runner = ClipRunner(model, seq, dev, ITERS)

    with LookupTimer() as lt, torch.no_grad():
        for _ in range(a.warmup):
            runner.step()
        torch.cuda.synchronize()
        tdist.barrier()
        lt.on = True
        t0 = time.perf_counter()
        for _ in range(a.steps):
            runner.step()
        torch.cuda.synchronize()
        tdist.barrier()
        elapsed = time.perf_counter() - t0
        lt.on = False
        roof = lt.result()

    elapsed = tdist.max_over_ranks(elapsed)
    total_pairs = a.steps * max(world, 1)
    value = total_pairs / elapsed

    # accuracy of the synthetic run (random-init weights: parity, not quality, is what is checked)
    gpu_preds = None
    cpu = None
    epe_vs_oracle = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from tcs_mi355.harness import run_sequence
        sub = type(seq)(seq.frames[:2], seq.K, seq.baseline)
        gpu_preds = []
        run_sequence(model, sub, iters=ITERS, device=dev, collect=gpu_preds)
        cpu, epes = cpu_baseline(W, seq, gpu_preds, 2)
        epe_vs_oracle = [round(e, 6) for e in epes]

    # the run's only collective: per-rank [frames, elapsed] (EPE statistics ride the same vector in eval runs)
    vecs = tdist.gather_vectors(np.array([a.steps, elapsed], np.float64))

    if rank == 0:
        line = {
            "metric": "stereo-pairs/sec at 640x480 D=192, 32 GRU iters", "value": round(value, 4), "unit": "stereo-pairs/s",
            "n_gpus": max(world, 1), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 640x480 synthetic sequence len=10, D=192, 32 iters, one sequence per GPU",
                       "frames_per_rank": a.steps, "weights": "key-seeded synthetic (tcs_mi355.weights)"},
            "roofline": roof, "cpu_baseline": cpu, "epe_vs_oracle_first_frames": epe_vs_oracle,
            "ranks_frames": [int(v[0]) for v in vecs],
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
