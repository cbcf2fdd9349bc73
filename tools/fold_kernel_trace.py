#!/usr/bin/env python3
"""Fold a rocprofv3 kernel-trace CSV per (kernel, grid, phase): calls, avg/min/max us, total ms.

usage: fold_kernel_trace.py <..._kernel_trace.csv> <out.csv> [skip_first_n_ms]

`phase` separates the two populations a kernel can belong to in a bench.py run:
  * "burst"  — the launch directly follows another launch of the SAME kernel on the same queue (bench.py's event-timed
               lookup bursts: 200 identical launches back to back; conv micro-benchmarks);
  * "loop"   — it follows a different kernel (the frame graph: every launch sits behind a dependent producer).
rocprofv3's per-kernel interval includes dispatch/completion overhead that differs between the two (a trivial kernel in
"loop" position reads 4.4-5 us), so the two must not be averaged together.  The column `gap_before_us` is the mean idle
time between the previous kernel's end and this kernel's start on the same queue: the dispatch floor a reader can
subtract."""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
skip_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = {}          # queue -> (kernel name, end timestamp)
agg = collections.OrderedDict()
for r in rows:
    start, end = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = (r.get("Agent_Id", ""), r.get("Queue_Id", ""))
    name = r["Kernel_Name"]
    p = prev.get(q)
    prev[q] = (name, end)
    if start - t0 < skip_ns:
        continue
    phase = "burst" if (p is not None and p[0] == name) else "loop"
    gap = (start - p[1]) / 1e3 if p is not None else 0.0
    d = (end - start) / 1e3
    k = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
         r.get("VGPR_Count", ""), phase)
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d); a[4] += max(gap, 0.0)
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_x", "wg_x", "vgpr", "phase", "calls", "avg_us", "min_us", "max_us", "total_ms", "gap_before_us"])
    for (name, g, wg, vg, phase), (n, tot, mn, mx, gap) in out:
        w.writerow([name[:120], g, wg, vg, phase, n, round(tot / n, 3), round(mn, 3), round(mx, 3), round(tot / 1e3, 3), round(gap / n, 3)])
print("kernels", len(out), "total ms", round(sum(v[1] for v in agg.values()) / 1e3, 2))
