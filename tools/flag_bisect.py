#!/usr/bin/env python3
"""Locates what sets the S16 domain flags (tcs_s16_flags) on BASELINE configs[1] (10-frame 640x480 clip, 32 iterations).

Phase 1: the clip as bench.py runs it (HIP-graph replay, parallel branches), flag words read per frame and per source file
         (tcs_s16_flags_detail), outputs checked for finiteness.
Phase 2: the same clip with eager launches.
Phase 3: for the first frame that trips, that frame is re-run eagerly from the saved temporal state with a flag read after
         EVERY library call (device-synchronising: the parallel branches then run one after another): first call that trips, per
         iteration, with the ranges of the tensors involved.

    python tools/flag_bisect.py [--frames 10] [--iters 32] [--seed 2000]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths  # noqa: E402

tcs_paths.add_product_path()
import torch  # noqa: E402

UNITS = ["conv_s16", "s16_ops", "conv(f32/7x7)", "conv_f16", "stencil"]


def detail():
    from tcs_mi355 import native as nv
    u = (C.c_uint * 5)()
    rc = nv.lib().tcs_s16_flags_detail(u)
    assert rc == 0, rc
    return list(u)


def fmt(u):
    return " ".join(f"{n}={v:#x}" for n, v in zip(UNITS, u) if v) or "clean"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--seed", type=int, default=2000)
    ap.add_argument("--passes", type=int, default=2)
    a = ap.parse_args()
    import bench
    from tcs_mi355 import native as nv
    from tcs_mi355 import synth
    dev = torch.device("cuda:0")
    model, W = bench.build_model(dev)
    seq = synth.make_sequence(a.seed, n_frames=a.frames, height=480, width=640, max_disp=192.0)

    def run_clip(label, graph, passes):
        model.use_hip_graph = graph
        runner = bench.ClipRunner(model, [seq], dev, a.iters)
        first = None
        states = {}
        detail()
        for ps in range(passes):
            for t in range(a.frames):
                states[t] = runner.state
                out = runner.step()
                u = detail()
                fin = all(bool(torch.isfinite(v).all()) for v in [out["flow"], out["flow_q"], *out["net_list"]])
                mx = float(out["flow_q"].abs().max())
                nmx = max(float(n.abs().max()) for n in out["net_list"])
                print(f"[{label}] pass {ps} frame {t}: flags {fmt(u)}; outputs finite={fin} max|flow_q|={mx:.2f} max|net|={nmx:.4f}", flush=True)
                if any(u) and first is None:
                    first = (t, states[t])
        return first

    with torch.no_grad():
        f_graph = run_clip("graph", True, a.passes)
        f_eager = run_clip("eager", False, 1)
        first = f_eager or f_graph
        if first is None:
            print("no flag tripped")
            return
        t, state = first
        print(f"--- per-call bisect of frame {t} (eager, flags read after every library call) ---", flush=True)
        model.use_hip_graph = False
        calls = []
        orig = nv.check

        def check(rc, what):
            orig(rc, what)
            u = detail()
            calls.append(what)
            if any(u):
                print(f"  call #{len(calls)} {what}: {fmt(u)}", flush=True)

        nv.check = check
        runner = bench.ClipRunner(model, [seq], dev, a.iters)
        runner.t, runner.state = t, state
        try:
            runner.step()                      # the frame's own schedule (one call at a time)
            print(f"  {len(calls)} library calls in the frame; now with the trace hook (states as of the end of each iteration)", flush=True)
            runner.t, runner.state = t, state
            calls.clear()
            trace = {}
            model._trace = trace
            runner.step()
        finally:
            nv.check = orig
            model._trace = None
        for i, it in enumerate(trace.get("iters", [])):
            rng = {k: float(v.abs().max()) for k, v in it.items() if torch.is_tensor(v)}
            nets = [float(n.abs().max()) for n in it["net"]]
            print(f"  iter {i}: " + " ".join(f"max|{k}|={v:.3g}" for k, v in rng.items()) + f" max|net|={nets}", flush=True)


if __name__ == "__main__":
    main()
