#!/bin/bash
# bench.py with the roofline pass but without CPU baseline / extra legs, per TCS_MI355_X variant; prints frame time and the lookup's in-frame clock.
# usage: tools/bench_roof.sh <out dir under gpurun_out> <variant> [<variant> ...]     ("-" = default)
out=gpurun_out/$1; shift; mkdir -p $out
for v in "$@"; do
  x=${v%%@*}; [ "$x" = "-" ] && x=""
  lib=""; [[ "$v" == *@* ]] && lib=$(pwd)/temporally-consistent-stereo-matching_amd/lib/libtcs_mi355_${v#*@}.so      # "tokens@libvariant"
  [ -n "$lib" ] && export TCS_MI355_LIB=$lib || unset TCS_MI355_LIB
  TCS_MI355_X=$x timeout -k 10 300 python bench.py --steps 20 --no-cpu-baseline --batched-leg 0 --drop-in-steps 0 --kitti-steps 0 2>/dev/null > $out/bench_$v.json || { echo "[$v] FAILED"; exit 1; }
  python - "$out/bench_$v.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"[{sys.argv[2]}]", d["ms_per_step"], d["step_ms_min_median_max"], "lookup in-frame us", r["avg_launch_us"], "frac", r["frac"], "min", r["min_launch_us"],
      "burst", r["burst_events"]["avg_launch_us"], "hot", (r.get("in_kernel_hot") or {}).get("median_launch_us"))
PY
done
