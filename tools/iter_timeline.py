#!/usr/bin/env python3
"""One steady-state refinement iteration out of a rocprofv3 kernel trace: queue, start/end (us, relative), duration, grid, kernel.
The iteration is cut between two consecutive launches of an anchor kernel (default k_hidden_update_s16).
usage: iter_timeline.py <kernel_trace.csv> [anchor-substring] [which-from-the-end]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_hidden_update_s16"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-back], idx[-back + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[max(a - 14, 0):b + 1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:48]
    print(f"q{r['Queue_Id']:>2s} {s:8.1f} {e:8.1f} {e - s:6.1f} g{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']:>3s} {n}")
