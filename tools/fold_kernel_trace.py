#!/usr/bin/env python3
"""Fold a rocprofv3 kernel trace CSV per (kernel, grid): calls, avg/min/max us, total ms.
usage: fold_kernel_trace.py <..._kernel_trace.csv> <out.csv> [skip_first_n_ms]"""
import csv, sys, collections
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
skip_ns = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.0
agg = collections.OrderedDict()
for r in rows:
    if int(r["Start_Timestamp"]) - t0 < skip_ns:
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
         r.get("VGPR_Count", ""))
    a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_x", "wg_x", "vgpr", "calls", "avg_us", "min_us", "max_us", "total_ms"])
    for (name, g, wg, vg), (n, tot, mn, mx) in out:
        w.writerow([name[:120], g, wg, vg, n, round(tot / n, 3), round(mn, 3), round(mx, 3), round(tot / 1e3, 3)])
print("kernels", len(out), "total ms", round(sum(v[1] for v in agg.values()) / 1e3, 2))
