#!/bin/bash
# Sweep of HIP-runtime environment settings over the end-to-end frame time, inside ONE gpurun call: `bench.py --quick --steps 20`
# per setting, each under its own time limit; the sweep STOPS at the first run that fails or times out (no GPU step after a kill).
# usage: tools/env_sweep.sh <rounds> "<VAR=val [VAR=val ...]>" ...      ("-" = no extra variable)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    envs=""; [ "$v" != "-" ] && envs="$v"
    out=$(env $envs timeout -k 10 150 python bench.py --steps 20 --warmup 3 --quick 2>/dev/null)
    rc=$?
    if [ $rc -ne 0 ]; then echo "round $r [$v] FAILED rc=$rc: stopping the sweep"; exit 1; fi
    echo "round $r [$v] $(echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_min_median_max'])")"
  done
done
