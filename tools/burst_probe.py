#!/usr/bin/env python3
"""What happens on the box during a slow frame?  Replays the temporal frame graph for N frames while a background thread samples
the GPU's sysfs telemetry (shader clock level, socket power, busy %) every few ms; prints per-frame GPU time next to the telemetry
seen during that frame.  (GPU box.)  usage: burst_probe.py [frames]"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120


def find(pattern):
    for p in sorted(glob.glob(pattern)):
        try:
            open(p).read()
            return p
        except OSError:
            continue
    return None


SCLK = find("/sys/class/drm/card*/device/pp_dpm_sclk")
POWER = find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") or find("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
BUSY = find("/sys/class/drm/card*/device/gpu_busy_percent")
print("telemetry files:", SCLK, POWER, BUSY)
samples, stop = [], False


def read(p):
    try:
        return open(p).read()
    except Exception:
        return ""


def sampler():
    while not stop:
        t = time.perf_counter()
        sclk = ""
        for line in read(SCLK).splitlines() if SCLK else []:
            if line.strip().endswith("*"):
                sclk = line.split(":")[1].replace("*", "").strip()
        pw = read(POWER).strip() if POWER else ""
        samples.append((t, sclk, int(pw) / 1e6 if pw.isdigit() else -1.0, read(BUSY).strip() if BUSY else ""))
        time.sleep(0.004)


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
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(frames + 1)]
    host = []
    for i in range(frames):
        ev[i].record()
        host.append(time.perf_counter())
        runner.step()
        if i % 8 == 7:
            torch.cuda.synchronize()          # keeps host time stamps close to GPU frame boundaries
    ev[frames].record()
    torch.cuda.synchronize()
    host.append(time.perf_counter())
    stop = True
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(frames)]
med = sorted(ms)[len(ms) // 2]
print(f"{frames} frames: median {med:.2f} ms, max {max(ms):.2f} ms, frames above 1.3 x median: {sum(m > 1.3 * med for m in ms)}")
for i, m in enumerate(ms):
    window = [s for s in samples if host[i] - 0.005 <= s[0] <= host[i + 1] + 0.03]
    clocks = sorted(set(s[1] for s in window))
    pw = [s[2] for s in window if s[2] >= 0]
    flag = "  <-- slow" if m > 1.3 * med else ""
    if flag or i % 10 == 0:
        print(f"frame {i:3d}: {m:6.2f} ms  sclk {clocks}  power {min(pw) if pw else -1:.0f}-{max(pw) if pw else -1:.0f} W  busy {sorted(set(s[3] for s in window))}{flag}")
