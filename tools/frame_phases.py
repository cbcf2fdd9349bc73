#!/usr/bin/env python3
"""Per-frame phases out of a rocprofv3 kernel trace of bench.py: where the time outside the refinement loop goes.
A frame ends with k_convex_upsample; its loop spans from the first k_corr_lookup<..> 'loop'-position launch after the previous frame's end
to the last conv1x1 blend epilogue / k_hidden_update before the upsample.  Prints, for the last N steady-state frames: frame period (end to end),
head (previous end -> first lookup of the loop), loop (first lookup -> last hidden-state update end), tail (-> upsample end), and how much
extract-stage work (k_conv7x7<3> .. k_corr_finalize) overlapped the previous frame's loop (prefetch).
usage: frame_phases.py <kernel_trace.csv> [N]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ends = [i for i, e in enumerate(ev) if "k_convex_upsample" in e[2]]
out = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = ev[a + 1:b + 1]
    hus = [e for e in seg if "k_hidden_update" in e[2]]
    looks = [e for e in seg if "k_corr_lookup" in e[2]]
    if len(hus) != 32 or len(looks) < 32:
        continue                                       # not a plain 32-iteration frame (captures, bursts)
    t_prev_end, t_end = ev[a][1], ev[b][1]
    t_loop0 = looks[0][0]
    t_loop1 = hus[-1][1]
    stem = [e for e in seg if "k_conv7x7<3>" in e[2]]
    out.append(((t_end - t_prev_end) / 1e6, (t_loop0 - t_prev_end) / 1e6, (t_loop1 - t_loop0) / 1e6, (t_end - t_loop1) / 1e6,
                (stem[0][0] - t_prev_end) / 1e6 if stem else float("nan")))
print("frame_ms  head_ms  loop_ms  tail_ms  rgb_stem_start_after_prev_end_ms   (last %d plain frames of %d)" % (min(N, len(out)), len(out)))
for o in out[-N:]:
    print("%8.2f %8.2f %8.2f %8.2f %8.2f" % o)
# the head of the last plain frame, launch by launch (queue, start / end in us after the previous frame's end, duration, grid, kernel)
import re
plain = []
for a, b in zip(ends[:-1], ends[1:]):
    seg = rows_sorted = None
    n_hu = sum(1 for e in ev[a + 1:b + 1] if "k_hidden_update" in e[2])
    if n_hu == 32:
        plain.append((a, b))
if plain:
    a, b = plain[-1]
    t0 = ev[a][1]
    print("\nhead of the last plain frame:")
    for i in range(a + 1, b + 1):
        r = rows[i]
        if "k_hidden_update" in r["Kernel_Name"]:
            break
        st, en = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:56]
        print(f"q{r['Queue_Id']:>2s} {st:8.1f} {en:8.1f} {en - st:6.1f} g{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d}x{r['Workgroup_Size_X']:>3s} {n}")
