#!/usr/bin/env python3
"""bench.py — stereo-pairs/sec of the TC-Stereo hot path on MI355X.

Metric (BASELINE.json): stereo-pairs/sec at 640x480, D=192, 32 GRU iterations.
A "step" is one stereo pair of a synthetic 10-frame 640x480 sequence through `TCStereo.forward`
(frame 0 takes the argmax branch, frames 1-9 the temporal-warp branch; state resets when the clip
wraps).  Inputs are resident in HBM before the timed region.  One process per GPU, one independent
sequence per rank (no data-path collective; one all_gather of EPE statistics at the end).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: spawns N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 with `roofline` (corr lookup: device-clock stamps of the frame's own launches, taken in a
stamped pass after the timed region), `cpu_baseline` (the CPU oracle on the host cores, rank 0 / N=1 only) and `domain_flags`
(the device-side NaN / saturation guard of the timed frames; non-zero = no headline, exit code 3).
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


CONTRACTION = ("convolutions: fp16 hi/lo split operands, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate (error <= fp32 MFMA "
               "chain; tests/test_gpu_parity.py::test_f16x3_split_is_fp32_grade), loop and frame layers alike; the gradient-candidate stem "
               "and the correlation volume: fp32 MFMA (v_mfma_f32_32x32x2_f32); everything else fp32 VALU")


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


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
    """Steps through device-resident clips, carrying the temporal state like evaluate_stereo.py:170-197.  Several
    independent sequences can ride the batch dimension (--seqs-per-gpu); one step = one frame of each."""

    def __init__(self, model, seqs, dev, iters, prefetch=True):
        from tcs_mi355.harness import InputPadder
        seqs = list(seqs) if isinstance(seqs, (list, tuple)) else [seqs]
        self.model, self.iters, self.n = model, iters, len(seqs[0].frames)
        K_raw = torch.stack([torch.as_tensor(q.K) for q in seqs]).to(dev)
        self.baseline = torch.tensor([q.baseline for q in seqs], device=dev)
        self.frames = []
        for t in range(self.n):
            i1 = torch.stack([torch.as_tensor(q.frames[t].image1) for q in seqs]).to(dev)
            i2 = torch.stack([torch.as_tensor(q.frames[t].image2) for q in seqs]).to(dev)
            padder = InputPadder(i1.shape, divis_by=32)
            (i1, i2), K = padder.pad(i1, i2, K=K_raw)
            T = torch.stack([torch.as_tensor(q.frames[t].T) for q in seqs]).to(dev)
            self.frames.append((i1.contiguous(), i2.contiguous(), K, T))
        self.t = 0
        self.state = None
        self.last = None
        self.prefetch = prefetch

    def step(self):
        i1, i2, K, T = self.frames[self.t]
        params = None
        if self.t > 0 and self.state is not None:
            flow_q, nets, fmap1, prev_T = self.state
            params = dict(K=K, T=T, previous_T=prev_T, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=self.baseline)
        out = self.model(i1, i2, iters=self.iters, test_mode=True, params=params)
        if self.prefetch:
            # a video loop knows its next frame (resident here): its image-only stage (features, correlation pyramid, context) is launched
            # right behind this frame and starts when this frame's refinement loop starts (TCStereo.prefetch); the clip wraps, and a
            # wrapped frame starts a new sequence
            nxt = (self.t + 1) % self.n
            self.model.prefetch(self.frames[nxt][0], self.frames[nxt][1], first=(nxt == 0), inputs_ready=True)
        self.state = (out["flow_q"], out["net_list"], out["fmap1"], T)
        self.last = out
        self.t = (self.t + 1) % self.n
        if self.t == 0:
            self.state = None
        return out


def _quarter(n):
    """1/4-resolution size of a frame dimension after InputPadder(divis_by=32)."""
    return (n + 31) // 32 * 32 // 4


def lookup_burst_us(dev, B, n=200, reps=5):
    """Average duration of one corr-lookup launch measured with HIP events on the launch stream: `n` back-to-back launches
    of the production kernel (same grid: B x 120 x 160 pixels, radius 4, 4 levels) are captured into a HIP graph like the
    frame's own launches, `reps` replays are bracketed by one event pair, and the elapsed time is divided by n * reps.
    A single launch cannot be bracketed: an event pair costs ~5 us on this stack, more than the kernel."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(77)
    h, w = _quarter(HEIGHT), _quarter(WIDTH)
    f1, f2 = torch.randn(B, 256, h, w, generator=gen).to(dev), torch.randn(B, 256, h, w, generator=gen).to(dev)
    pyr = ops.corr_build(f1, f2)
    xs = torch.arange(w, dtype=torch.float32).view(1, 1, 1, w).expand(B, 1, h, w)
    # a smooth disparity field (ramp + gentle waves), like the fields the refinement loop produces
    yy = torch.arange(h, dtype=torch.float32).view(1, 1, h, 1)
    disp = 4.0 + 30.0 * yy / h + 2.0 * torch.sin(xs / 17.0) * torch.cos(yy / 11.0)
    coords = (xs - disp).contiguous().to(dev)
    out = ops.corr_lookup(pyr, coords, 4)
    torch.cuda.synchronize()

    def burst(stamped):
        # `stamped`: every launch also records the device clock (first instruction of the first workgroup -> last acknowledged
        # store of the last one) into its own slot: the kernel's own span with the pyramid hot in L2 / Infinity Cache.  The
        # event-timed burst runs WITHOUT stamps (they add a wait for the store acknowledgement and an atomic per wave).
        saved, probe = ops.LOOKUP_PROBE, (ops.LookupProbe(dev, slots=n) if stamped else None)
        ops.LOOKUP_PROBE = probe
        try:
            g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            with torch.cuda.stream(side):
                with torch.cuda.graph(g, stream=side):
                    for _ in range(n):
                        ops.corr_lookup(pyr, coords, 4, out=out)
        finally:
            ops.LOOKUP_PROBE = saved
        g.replay()
        torch.cuda.synchronize()
        if probe is not None:
            probe.reset()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / (n * reps), (probe.durations_us() if probe is not None else None)

    events_us, _ = burst(False)
    _, hot = burst(True)
    return events_us, (float(np.median(hot)) if hot else None)


def lookup_roofline(probe, snapshots, burst, pixels):
    """`roofline` object for the corr lookup.  `achieved` / `frac` / `avg_launch_us` are the kernel's launches INSIDE the frame:
    the in-kernel device-clock interval (first instruction of the first workgroup to the last acknowledged store of the last one,
    s_memrealtime at 100 MHz) of every lookup launch of a stamped pass over the clip — the same graph as the timed frames plus the
    stamps.  That is the figure rocprofv3's kernel trace of this command reproduces (`rocprof_loop_avg_us`, `frac_rocprof`, from
    the committed profile; rocprofv3's interval adds its dispatch / completion overhead).  The isolated, cache-hot figures —
    HIP events around graph-replayed bursts of 200 launches, and the stamps of those burst launches — are reported beside it as
    `burst_events` and `in_kernel_hot`: they say what the kernel does alone, not what the frame gets."""
    burst_us, hot_us = burst
    durs = []
    for snap in snapshots:
        durs += probe.durations_us(snap)
    alg_bytes = LOOKUP_BYTES_PER_PIXEL * pixels
    rate = lambda us: round(alg_bytes / (us * 1e-6) / 1e9, 1)
    frac = lambda us: round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    dur_us = float(np.mean(durs)) if durs else float("nan")
    roof = {"bound": "hbm", "kernel": "k_corr_lookup<4>", "achieved": rate(dur_us) if durs else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": frac(dur_us) if durs else None, "traffic": None, "avg_launch_us": round(dur_us, 3) if durs else None,
            "min_launch_us": round(float(np.min(durs)), 3) if durs else None, "launches": len(durs),
            "algorithmic_bytes_per_launch": alg_bytes,
            "timer": "in-kernel s_memrealtime stamps (100 MHz) of every lookup launch of a stamped pass over the clip (frame graph replay)"}
    roof["burst_events"] = {"avg_launch_us": round(burst_us, 3), "achieved": rate(burst_us), "frac": frac(burst_us),
                            "timer": "HIP events around graph-replayed bursts of 200 back-to-back launches, kernel alone, caches hot"}
    if hot_us:
        roof["in_kernel_hot"] = {"median_launch_us": round(hot_us, 3), "achieved": rate(hot_us), "frac": frac(hot_us),
                                 "timer": "in-kernel stamps of the burst launches (pyramid and coordinates cache-hot)"}
    # PMC traffic and rocprofv3's own durations come from the committed profile of this same command (profiles/README.md):
    # bench.py cannot run the profiler on itself.  rocprofv3's per-kernel interval includes ~2 us of dispatch / completion
    # (trivial kernels read 4.5-5 us in the same trace), so frac_rocprof is a lower bound of the kernel's own rate.
    for name in ("r04_lookup_pmc.json", "r03_lookup_pmc.json", "r02_lookup_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
            rows = list(pmc.get("workloads", {}).items()) or [("", pmc)]
            hit = next(((k, r) for k, r in rows if r.get("algorithmic_bytes_per_launch") == alg_bytes), None)
            if hit is None:
                continue
            key, row = hit
            if pmc.get("algorithmic_bytes_per_launch") == alg_bytes and "traffic_bytes_per_launch" in pmc:
                roof["traffic"] = pmc["traffic_bytes_per_launch"]
                roof["traffic_source"] = f"profiles/{name} (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate --pmc passes)"
            for k in ("rocprof_burst_avg_us", "rocprof_loop_avg_us", "rocprof_kernel_trace_avg_us"):
                if k in row:
                    roof[k] = row[k]
            rp = row.get("rocprof_loop_avg_us") or row.get("rocprof_burst_avg_us") or row.get("rocprof_kernel_trace_avg_us")
            if rp:
                roof["frac_rocprof"] = frac(rp)
                roof["frac_rocprof_source"] = (f"profiles/{name}" + (f" [{key}]" if key else "") +
                                               ": rocprofv3 --kernel-trace average of the frame's own lookup launches")
            break
        except (OSError, KeyError, ValueError):
            continue
    return roof


def cpu_baseline(W, seq, gpu_preds, n_frames=3):
    """The CPU oracle ("port" of the reference's fp32 CPU path) on the first frames of the same clip: one untimed
    warm-up frame (thread pool, allocator, first-touch), then `n_frames` frames timed one by one; the value is
    1 / median frame time (1 warm-up + median of >= 3)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tcs_oracle as oracle
    from tcs_mi355.harness import run_sequence
    # the GPU box exposes every host core but grants a 16-core share: more threads than that only thrash
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, os.cpu_count() or 1, int(os.environ.get("TCS_BENCH_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    log(f"cpu baseline: oracle on {cores} threads, 1 warm-up + {n_frames} timed frames ...")
    stamps = []

    def fwd(a, b, **kw):
        out = oracle.tc_stereo_forward(W, a, b, iters=kw["iters"], params=kw["params"])
        stamps.append(time.perf_counter())
        return out

    warm = type(seq)(seq.frames[:1], seq.K, seq.baseline)
    run_sequence(fwd, warm, iters=ITERS, device=torch.device("cpu"))
    sub = type(seq)(seq.frames[:n_frames], seq.K, seq.baseline)
    preds = []
    stamps.clear()
    t0 = time.perf_counter()
    run_sequence(fwd, sub, iters=ITERS, device=torch.device("cpu"), collect=preds)
    times = np.diff(np.array([t0] + stamps))
    med = float(np.median(times))
    epes = [float((g.cpu() - p).abs().mean()) for g, p in zip(gpu_preds, preds)]
    return {"value": round(1.0 / med, 4), "unit": "stereo-pairs/s", "cores": cores, "kind": "port",
            "frame_seconds": [round(float(t), 3) for t in times],
            "sample": f"median of the first {n_frames} frames of the same 640x480 clip after 1 untimed warm-up frame, 32 iters, "
                      f"oracle/tcs_oracle.py on torch CPU fp32, {torch.get_num_threads()} threads"}, epes


def real_data_leg(a, dev):
    """BASELINE configs[2] (TartanAir abandonedfactory/Easy/P000, 480x640, pretrained weights, 32 iters): EPE / D1 and
    pairs/s through the evaluation harness.  Neither the dataset nor the checkpoint ships with the repository (the
    reference links them externally, README.md:36-82): when either is missing the leg is skipped and says why."""
    from tcs_mi355 import harness
    if not os.path.exists(a.ckpt):
        return {"status": "skipped", "reason": f"checkpoint {a.ckpt!r} not found (no pretrained weights offline)"}
    seq = harness.load_tartanair_sequence(a.tartanair, max_frames=30)
    if seq is None:
        return {"status": "skipped", "reason": f"TartanAir folder: {harness.load_tartanair_sequence.why}"}
    try:
        from argparse import Namespace

        from core.tc_stereo import TCStereo
        args = Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2,
                         context_norm="none", slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
        model = TCStereo(args)
        n = harness.load_checkpoint(model, a.ckpt)
        model = model.to(dev).eval()
        harness.run_sequence(model, type(seq)(seq.frames[:2], seq.K, seq.baseline), iters=ITERS, device=dev)      # captures
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = harness.run_sequence(model, seq, iters=ITERS, device=dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red = harness.reduce_stats([stats.vector()])
        return {"status": "ran", "frames": len(seq.frames), "tensors_loaded": n, "pairs_per_s_incl_host_io": round(len(seq.frames) / dt, 3),
                "epe": round(red["epe"], 4), "d1": round(red["d1"], 3), "d3": round(red["d3"], 3)}
    except Exception as e:
        return {"status": "failed", "reason": f"{type(e).__name__}: {e}"}


def count_gpus_sysfs(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs the kernel driver exposes, read from the KFD topology (nodes with simd_count > 0; CPUs have 0) — no HIP, no amdsmi, so the
    launcher process stays GPU-free.  Honours ROCR/HIP_VISIBLE_DEVICES by taking the shorter list.  None when the topology is unreadable
    (then nothing is checked here: a rank whose LOCAL_RANK has no device fails fast and takes the others down)."""
    import glob
    n = 0
    files = glob.glob(os.path.join(root, "*", "properties"))
    if not files:
        return None
    try:
        for f in files:
            with open(f) as fh:
                for line in fh:
                    k, _, v = line.partition(" ")
                    if k == "simd_count" and int(v) > 0:
                        n += 1
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([t for t in os.environ[var].split(",") if t.strip()]))
    return n


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (one per GPU, RCCL over xGMI) BEFORE this
    process has made any GPU call, relay rank 0's JSON line, exit with the worst return code.  The parent never touches
    the GPU (a process that has initialised HIP must not exec or fork GPU workers on this pool)."""
    import socket
    import subprocess
    if not a.dry_run and os.environ.get("TCS_MI355_DIST_BACKEND") != "gloo":      # (gloo rehearsal: ranks may share a GPU)
        have = count_gpus_sysfs()                    # no HIP / amdsmi call in the parent: it only starts the ranks
        if have is not None and have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: this box exposes {have} GPU(s); one rank per GPU is the only supported layout")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies (bad device, import error) must not leave the others waiting in the rendezvous for its time-out
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()                    # exactly the processes started above
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.write("".join(out0))
    sys.stdout.flush()
    if failed or any(rcs):
        raise SystemExit(f"rank return codes {rcs}")


def dry_run(a, tdist, rank, world):
    """The multi-rank plumbing of main() with the model replaced by a sleep: what the world-size-2 gloo test drives."""
    import torch.distributed as dist
    tdist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    tdist.barrier()
    elapsed = time.perf_counter() - t0
    vecs = tdist.gather_vectors(np.array([a.steps, elapsed], np.float64))
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_frames": [int(v[0]) for v in vecs],
                          "dist_world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "dist_backend": dist.get_backend() if dist.is_initialized() else None,
                          "value": sum(int(v[0]) for v in vecs) / max(float(v[1]) for v in vecs)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60,
                    help="timed frames (default 60 = six passes over the 10-frame clip, ~1.7 s: long enough that the box's occasional slow "
                         "bursts (DESIGN.md section 6) enter the mean with their average weight instead of by luck)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch kernels eagerly instead of replaying HIP graphs")
    ap.add_argument("--prefetch", action="store_true",
                    help="call TCStereo.prefetch(next frame) after every forward(): the optional video-loop extension (enqueues the next frame's "
                         "image-only stage ahead of time).  Default OFF since round 4: the headline is exactly the calls evaluate_stereo.py:170-197 makes")
    ap.add_argument("--no-prefetch", action="store_true", help="(the default; accepted for round-3 command lines)")
    ap.add_argument("--seqs-per-gpu", type=int, default=1,
                    help="independent sequences stacked on the batch dimension of every launch (default 1 = BASELINE configs[1])")
    ap.add_argument("--size", default=None, metavar="HxW",
                    help="frame size other than BASELINE configs[1]'s 480x640, e.g. 375x1242 for configs[4] (KITTI raw latency)")
    ap.add_argument("--batched-leg", type=int, default=4,
                    help="after the timed run, also time this many sequences per launch (reported under batched_leg; 0/1 = skip)")
    ap.add_argument("--drop-in-steps", type=int, default=20,
                    help="frames of the prefetch leg (the headline's clip with TCStereo.prefetch after every forward; reported under prefetch_leg; 0 = skip)")
    ap.add_argument("--kitti-steps", type=int, default=10,
                    help="frames of the KITTI-shape latency leg (375x1242, BASELINE configs[4]; reported under kitti_leg; 0 = skip)")
    ap.add_argument("--tartanair", default=os.environ.get("TCS_TARTANAIR_SEQ", "datasets/TartanAir/abandonedfactory/Easy/P000"),
                    help="BASELINE configs[2]: a TartanAir trajectory folder (image_left/, image_right/, depth_left/, pose_left.txt); "
                         "skipped with a logged reason when absent")
    ap.add_argument("--ckpt", default=os.environ.get("TCS_CKPT", "checkpoints/tartanair.pth"),
                    help="reference checkpoint (.pth with a 'model' state dict) for the TartanAir leg; weights-only load")
    ap.add_argument("--quick", action="store_true",
                    help="A/B runs (tools/ab_bench.sh): the timed region, the domain flags and the evaluation gather only — no roofline pass, "
                         "no CPU baseline, no extra legs")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: ranks, rendezvous, barrier and the statistics gather only (tests)")
    a = ap.parse_args()

    if a.quick:
        a.no_cpu_baseline, a.batched_leg, a.drop_in_steps, a.kitti_steps = True, 0, 0, 0
    a.no_prefetch = not a.prefetch
    global HEIGHT, WIDTH
    if a.size:
        HEIGHT, WIDTH = (int(v) for v in a.size.lower().split("x"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a, sys.argv[1:])           # before anything touches the GPU
    from tcs_mi355 import dist as tdist
    rank, world, local = tdist.init_from_env("gloo" if a.dry_run else None)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torchrun --nproc-per-node {a.gpus}, or without "
                         f"a torchrun environment (bench.py then starts the ranks itself)")
    if a.dry_run:
        return dry_run(a, tdist, rank, world)
    from tcs_mi355 import native, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if not a.eager:
        os.environ["TCS_MI355_GRAPH_STRICT"] = "1"   # a failed capture must fail the run, not degrade it to eager launches
    native.lib()
    if os.environ.get("TCS_MI355_DIST_BACKEND") == "gloo":        # rehearsal: several ranks may share a GPU
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    log("building model + synthetic clip")
    model, W = build_model(dev)
    model.use_hip_graph = not a.eager
    S = max(1, a.seqs_per_gpu)
    seqs = [synth.make_sequence(2000 + rank * S + j, n_frames=CLIP_LEN, height=HEIGHT, width=WIDTH, max_disp=MAX_DISP) for j in range(S)]
    seq = seqs[0]
    runner = ClipRunner(model, seqs, dev, ITERS, prefetch=not a.no_prefetch)

    from tcs_mi355 import ops, s16
    with torch.no_grad():
        log("warm-up (captures the HIP graphs)")
        for _ in range(max(a.warmup, 2)):          # >= 2 so that both branches (first frame / temporal) are captured
            runner.step()
        s16.take_flags()                           # clear: the flags below are those of the timed frames (synchronises)
        log(f"timing {a.steps} steps")
        tdist.barrier()
        t0 = time.perf_counter()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        for i in range(a.steps):
            marks[i].record()                      # per-step spread (diagnostic): events only, no host sync inside the region
            runner.step()                          # the production graph: no stamps, no copies
        marks[-1].record()
        torch.cuda.synchronize()
        tdist.barrier()
        elapsed = time.perf_counter() - t0
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    # the device-side replacement of the reference's NaN asserts (update.py:27-35,58-67,78-86,155-158): a timed frame whose
    # activations left the S16 domain (bit 0: clamped at 65504, bit 1: NaN / Inf) was timed on garbage — no headline for it
    domain_flags = s16.take_flags()
    graphs = getattr(model, "_graphs", None)
    if not a.eager and (graphs is None or graphs.fell_back or any(v is None for v in graphs.cache.values())):
        raise SystemExit("bench.py: a frame ran with eager launches although HIP-graph replay was requested "
                         f"(fell_back={getattr(graphs, 'fell_back', None)}); refusing to report it as graph replay")
    # per-rank evaluation statistics of the same clip against the synthetic ground truth, through the harness: the vector every
    # rank contributes to the run's one collective (evaluate_stereo.py:202-220; BASELINE configs[3]'s "RCCL EPE gather")
    from tcs_mi355.harness import reduce_stats, run_sequence
    with torch.no_grad():
        eval_stats = run_sequence(model, seq, iters=ITERS, device=dev)
    domain_flags |= eval_stats.domain_flags

    # roofline pass: the same clip with the lookup's device-clock stamps on (graphs re-captured with the stamp slots baked in)
    roof = None
    with torch.no_grad():
        try:
            if rank != 0 or a.quick:
                raise StopIteration                # the roofline object is rank 0's
            probe = ops.LookupProbe(dev, slots=64)
            ops.LOOKUP_PROBE = probe
            model._pipeline().drop()
            runner_r = ClipRunner(model, seqs, dev, ITERS, prefetch=not a.no_prefetch)
            snaps = []
            for _ in range(2):
                runner_r.step()
            probe.reset()
            for _ in range(CLIP_LEN):
                runner_r.step()
                snaps.append(probe.buf.clone())    # stream-ordered 300 KB copy + clear; no host sync
                probe.reset()
            torch.cuda.synchronize()
            ops.LOOKUP_PROBE = None
            model._pipeline().drop()               # later legs capture without stamps again
            roof = lookup_roofline(probe, snaps, lookup_burst_us(dev, S), S * _quarter(HEIGHT) * _quarter(WIDTH))
        except StopIteration:
            pass
        except Exception as e:                    # never lose the headline line over the auxiliary timing
            log(f"lookup roofline timing failed: {type(e).__name__}: {e}")
        finally:
            ops.LOOKUP_PROBE = None

    # the run's only collective (besides the two barriers): per-rank [pairs, elapsed]; the MAX over ranks of the elapsed
    # time and the aggregate come from it (EPE statistics ride the same vector in evaluation runs)
    vecs = tdist.gather_vectors(np.concatenate([np.array([a.steps * S, elapsed, float(domain_flags)], np.float64), eval_stats.vector()]))
    elapsed = max(float(v[1]) for v in vecs)
    total_pairs = sum(int(v[0]) for v in vecs)
    value = total_pairs / elapsed
    all_flags = 0
    for v in vecs:
        all_flags |= int(v[2])
    gathered_eval = reduce_stats([v[3:] for v in vecs])
    if all_flags:
        log(f"S16 domain flags {all_flags:#x} on the timed frames (bit 0: an activation was clamped at 65504, bit 1: NaN / Inf): the "
            f"frames were timed on saturated tensors; refusing to print a headline")
        raise SystemExit(3)

    # accuracy of the synthetic run (random-init weights: parity, not quality, is what is checked)
    gpu_preds = None
    cpu = None
    epe_vs_oracle = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            from tcs_mi355.harness import run_sequence
            sub = type(seq)(seq.frames[:3], seq.K, seq.baseline)
            gpu_preds = []
            log("accuracy sample on the GPU")
            run_sequence(model, sub, iters=ITERS, device=dev, collect=gpu_preds)
            cpu, epes = cpu_baseline(W, seq, gpu_preds, 3)
            epe_vs_oracle = [round(e, 6) for e in epes]
        except Exception as e:                    # report the failure, keep the headline line
            log(f"cpu baseline leg failed: {type(e).__name__}: {e}")
            cpu = {"value": None, "unit": "stereo-pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"}

    # extra leg (not `value`): the same clip with several independent sequences per launch.  One 640x480 sequence leaves
    # most launches under-filled (300-600 workgroups, 5.9 MB per lookup); this shows what the kernels do when fed.
    batched = None
    if rank == 0 and world == 1 and S == 1 and a.batched_leg > 1:
      try:
          Sb = a.batched_leg
          log(f"batched leg: {Sb} sequences per launch")
          seqs_b = [seq] + [synth.make_sequence(2000 + j, n_frames=CLIP_LEN, height=HEIGHT, width=WIDTH, max_disp=MAX_DISP)
                            for j in range(1, Sb)]
          runner_b = ClipRunner(model, seqs_b, dev, ITERS, prefetch=not a.no_prefetch)
          with torch.no_grad():
              for _ in range(2):
                  runner_b.step()
              torch.cuda.synchronize()
              tb = time.perf_counter()
              for _ in range(a.steps):
                  runner_b.step()
              torch.cuda.synchronize()
              tb = time.perf_counter() - tb
              flags_b = s16.take_flags()
              # stamped pass for the lookup's in-frame duration at this batch size
              ops.LOOKUP_PROBE = probe_b = ops.LookupProbe(dev, slots=64)
              model._pipeline().drop()
              runner_s = ClipRunner(model, seqs_b, dev, ITERS, prefetch=not a.no_prefetch)
              snaps_b = []
              for _ in range(2):
                  runner_s.step()
              probe_b.reset()
              for _ in range(CLIP_LEN):
                  runner_s.step()
                  snaps_b.append(probe_b.buf.clone())
                  probe_b.reset()
              torch.cuda.synchronize()
              ops.LOOKUP_PROBE = None
              model._pipeline().drop()
              roof_b = lookup_roofline(probe_b, snaps_b, lookup_burst_us(dev, Sb), Sb * _quarter(HEIGHT) * _quarter(WIDTH))
          batched = {"seqs_per_gpu": Sb, "value": round(a.steps * Sb / tb, 4), "unit": "stereo-pairs/s",
                     "ms_per_step": round(1e3 * tb / a.steps, 3), "domain_flags": flags_b,
                     "lookup": {k: roof_b[k] for k in ("achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch", "burst_events")}}
          del runner_b, runner_s
      except Exception as e:                      # the extra leg must never cost the headline line
        log(f"batched leg failed: {type(e).__name__}: {e}")
        batched = {"seqs_per_gpu": a.batched_leg, "error": f"{type(e).__name__}: {e}"}
        ops.LOOKUP_PROBE = None

    # extra leg: the same clip WITH the optional TCStereo.prefetch call after every forward() (the headline is without it)
    drop_in = None
    if rank == 0 and world == 1 and S == 1 and a.no_prefetch and a.drop_in_steps > 0:
        try:
            log("prefetch leg: TCStereo.prefetch(next frame) after every forward()")
            runner_d = ClipRunner(model, seqs, dev, ITERS, prefetch=True)
            with torch.no_grad():
                for _ in range(2):
                    runner_d.step()
                torch.cuda.synchronize()
                td = time.perf_counter()
                for _ in range(a.drop_in_steps):
                    runner_d.step()
                torch.cuda.synchronize()
                td = time.perf_counter() - td
            drop_in = {"value": round(a.drop_in_steps / td, 4), "unit": "stereo-pairs/s", "ms_per_step": round(1e3 * td / a.drop_in_steps, 3),
                       "steps": a.drop_in_steps, "domain_flags": s16.take_flags(),
                       "what": "as the headline plus TCStereo.prefetch(next images) after every forward() (optional video-loop extension: enqueues the "
                               "next frame's image-only stage ahead of time; hides host launch work only)"}
            del runner_d
        except Exception as e:
            log(f"prefetch leg failed: {type(e).__name__}: {e}")
            drop_in = {"error": f"{type(e).__name__}: {e}"}

    # extra leg: BASELINE configs[4], KITTI-raw frame shape (1242x375 -> padded 1248x384), 32 iterations, per-frame LATENCY: every
    # frame is bracketed by a device synchronisation (evaluate_stereo.py:85-89 times frames like that).  Captured once, with the
    # lookup's stamps on (one atomic per wave and a store acknowledgement per lookup launch: < 0.2 % of a frame), so the same pass
    # gives the lookup's in-frame duration at this size.
    kitti = None
    if rank == 0 and world == 1 and S == 1 and (HEIGHT, WIDTH) == (480, 640) and a.kitti_steps > 0:
        try:
            log("KITTI-shape leg: 375x1242, per-frame latency")
            kh, kw = 375, 1242
            seq_k = synth.make_sequence(2100, n_frames=CLIP_LEN, height=kh, width=kw, max_disp=MAX_DISP, K=synth.KITTI_K, baseline=0.54)
            ops.LOOKUP_PROBE = probe_k = ops.LookupProbe(dev, slots=64)
            model._pipeline().drop()
            runner_k = ClipRunner(model, [seq_k], dev, ITERS, prefetch=False)
            lat, snaps_k = [], []
            with torch.no_grad():
                for _ in range(2):
                    runner_k.step()
                probe_k.reset()
                torch.cuda.synchronize()
                for _ in range(a.kitti_steps):
                    tk = time.perf_counter()
                    runner_k.step()
                    torch.cuda.synchronize()
                    lat.append(1e3 * (time.perf_counter() - tk))
                    snaps_k.append(probe_k.buf.clone())
                    probe_k.reset()
                torch.cuda.synchronize()
            ops.LOOKUP_PROBE = None
            model._pipeline().drop()
            px_k = _quarter(kh) * _quarter(kw)
            durs_k = [d for sn in snaps_k for d in probe_k.durations_us(sn)]
            lk = float(np.mean(durs_k)) if durs_k else None
            lat.sort()
            kitti = {"size": "1242x375 (padded 1248x384)", "iters": ITERS, "frames": a.kitti_steps,
                     "latency_ms_min_median_max": [round(lat[0], 3), round(lat[len(lat) // 2], 3), round(lat[-1], 3)],
                     "pairs_per_s_from_median_latency": round(1e3 / lat[len(lat) // 2], 3), "domain_flags": s16.take_flags(),
                     "lookup": {"algorithmic_bytes_per_launch": LOOKUP_BYTES_PER_PIXEL * px_k, "avg_launch_us": None if lk is None else round(lk, 3),
                                "frac": None if lk is None else round(LOOKUP_BYTES_PER_PIXEL * px_k / (lk * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                "timer": "in-kernel stamps of the frame's own launches"},
                     "what": "BASELINE configs[4] shape, synthetic frames and key-seeded weights (no KITTI data or checkpoint offline); "
                             "host clock around forward() + device synchronisation per frame, lookup stamps on"}
            del runner_k
        except Exception as e:
            log(f"KITTI leg failed: {type(e).__name__}: {e}")
            kitti = {"error": f"{type(e).__name__}: {e}"}
            ops.LOOKUP_PROBE = None


    # BASELINE configs[2]: real TartanAir frames + the reference's pretrained weights, only when both are on the box
    real = None
    if rank == 0 and world == 1:
        real = real_data_leg(a, dev)

    if rank == 0:
        line = {
            "metric": f"stereo-pairs/sec at {WIDTH}x{HEIGHT} D=192, 32 GRU iters", "value": round(value, 4), "unit": "stereo-pairs/s",
            "n_gpus": max(world, 1), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
            "step_ms_min_median_max": [round(per_step[0], 3), round(per_step[len(per_step) // 2], 3), round(per_step[-1], 3)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "contraction": CONTRACTION,
            "config": {"workload": ("BASELINE configs[1]: 640x480" if (HEIGHT, WIDTH) == (480, 640) else f"{WIDTH}x{HEIGHT}")
                                   + " synthetic sequence len=10, D=192, 32 iters, "
                                   + ("one sequence per GPU" if S == 1 else f"{S} independent sequences batched per GPU"),
                       "frames_per_rank": a.steps * S, "seqs_per_gpu": S, "weights": "key-seeded synthetic (tcs_mi355.weights)",
                       "launch": ("eager" if a.eager else "hip-graph replay") + ("; forward() per frame only, the calls of evaluate_stereo.py:170-197 "
                                  "(prefetch_leg: the same with TCStereo.prefetch after every forward)" if a.no_prefetch else
                                  "; TCStereo.prefetch(next frame) after every forward(): the next frame's image-only stage is enqueued ahead of time "
                                  "(hides its host-side launch work; on the GPU it does not overlap the loop — DESIGN.md section 6)")},
            "domain_flags": all_flags,
            "gathered_eval_vs_synthetic_gt": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in gathered_eval.items()},
            "roofline": roof, "cpu_baseline": cpu, "epe_vs_oracle_first_frames": epe_vs_oracle, "batched_leg": batched,
            "prefetch_leg": drop_in, "kitti_leg": kitti, "tartanair_leg": real,
            "ranks_frames": [int(v[0]) for v in vecs],
            "dist_world_size": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "dist_backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
        }
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
