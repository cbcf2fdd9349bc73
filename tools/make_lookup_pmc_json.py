#!/usr/bin/env python3
"""profiles/rNN_lookup_pmc.json: the corr lookup's HBM traffic (two --pmc passes: FETCH_SIZE, WRITE_SIZE) and its rocprofv3 durations per
workload, from per-phase folds of kernel traces (tools/fold_kernel_trace.py).

usage: make_lookup_pmc_json.py --fetch <pmc dir> --write <pmc dir> --row <name>:<fold.csv>:<pixels per launch> [--row ...] --out <json>

The FIRST row is the headline workload (BASELINE configs[1], one 640x480 sequence): its figures are also written at the top level, where
bench.py reads `traffic_bytes_per_launch`, `algorithmic_bytes_per_launch`, `rocprof_loop_avg_us` and `rocprof_burst_avg_us`."""
import argparse
import csv
import glob
import json

BYTES_PER_PIXEL = 308          # 4 levels x 10 taps x 4 B + 4 B coordinate + 36 x 4 B out
HBM_PEAK = 8.0e12


def pmc_mean(d, counter):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_corr_lookup" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def fold_row(path, pixels):
    rows = list(csv.DictReader(open(path)))
    # the lookup launches of THIS workload: one thread per (pixel, level), i.e. grid_x ~ 4 * pixels (padding to whole workgroups and
    # XCD multiples adds a few percent).  A trace also holds lookups of other sizes — the single-sequence evaluation pass of a
    # batched run — and round 3 took whichever row came last per phase: its "four sequences: 0.54" was the one-sequence kernel's
    # 5.5 us divided into four sequences' bytes (the batched kernel's own row said 9.25 us = 0.32).
    mine = [r for r in rows if "k_corr_lookup" in r["kernel"] and 4 * pixels <= int(r["grid_x"]) <= 4.2 * pixels + 4096]
    by = {}
    for r in mine:
        if r["phase"] not in by or int(r["calls"]) > int(by[r["phase"]]["calls"]):
            by[r["phase"]] = r
    triv = [r for r in rows if r["phase"] == "loop" and any(k in r["kernel"] for k in ("k_flow_taps_step_grads", "k_flow_step_grads", "k_in_apply"))]
    alg = BYTES_PER_PIXEL * pixels
    out = {"pixels_per_launch": pixels, "algorithmic_bytes_per_launch": alg}
    for ph in ("loop", "burst"):
        if ph in by:
            us = float(by[ph]["avg_us"])
            out[f"rocprof_{ph}_avg_us"] = us
            out[f"rocprof_{ph}_calls"] = int(by[ph]["calls"])
            out[f"rocprof_{ph}_gap_before_us"] = float(by[ph]["gap_before_us"])
            out[f"frac_rocprof_{ph}"] = round(alg / (us * 1e-6) / HBM_PEAK, 4)
            out[f"rocprof_{ph}_kernel"] = by[ph]["kernel"][:24] + f" grid_x={by[ph]['grid_x']}"
    out["rocprof_trivial_kernels_in_loop_position_us"] = {r["kernel"][:40]: float(r["avg_us"]) for r in triv}
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--fetch")
ap.add_argument("--write")
ap.add_argument("--row", action="append", required=True)
ap.add_argument("--keep-pmc-of", help="an earlier JSON whose PMC fields (FETCH/WRITE passes) are carried over when --fetch/--write are not given")
ap.add_argument("--out", required=True)
a = ap.parse_args()
rows = {}
for spec in a.row:
    name, path, pixels = spec.split(":")
    rows[name] = fold_row(path, int(pixels))
first = next(iter(rows.values()))
out = {"kernel": "k_corr_lookup<4, 1>", "workloads": rows}
out.update({k: first[k] for k in ("algorithmic_bytes_per_launch", "rocprof_loop_avg_us", "rocprof_loop_calls", "rocprof_burst_avg_us",
                                  "rocprof_burst_calls") if k in first})
if a.fetch and a.write:
    fetch, nf = pmc_mean(a.fetch, "FETCH_SIZE")
    write, nw = pmc_mean(a.write, "WRITE_SIZE")
    if fetch is not None and write is not None:
        out.update({"launches_profiled_pmc": [nf, nw], "FETCH_SIZE_KB_raw_mean": round(fetch, 1), "WRITE_SIZE_KB_mean": round(write, 1),
                    "fetch_bytes_corrected_x2": int(2 * fetch * 1024), "write_bytes": int(write * 1024),
                    "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024)})
if a.keep_pmc_of and not (a.fetch and a.write):
    old = json.load(open(a.keep_pmc_of))
    for k in ("launches_profiled_pmc", "FETCH_SIZE_KB_raw_mean", "WRITE_SIZE_KB_mean", "fetch_bytes_corrected_x2", "write_bytes", "traffic_bytes_per_launch"):
        if k in old:
            out[k] = old[k]
out["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; 4-B-per-lane loads are a width the guide calls "
               "uncalibrated, so the true read traffic lies between raw and corrected).  loop = launches inside the frame graph behind a "
               "different kernel, burst = launches that directly follow another lookup launch (bench.py's event-timed graph bursts); "
               "rocprofv3's interval includes dispatch / completion overhead: see the trivial kernels' readings in the same position.  "
               "frac_rocprof_* = algorithmic bytes / rocprofv3 average duration / 8 TB/s.")
json.dump(out, open(a.out, "w"), indent=1)
print(json.dumps(out, indent=1))
