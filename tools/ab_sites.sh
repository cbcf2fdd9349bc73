#!/bin/bash
# A/B of fork sites inside ONE gpurun call: bench.py with TCS_MI355_FORK_SITES = all, and all minus each listed site.
# usage: tools/ab_sites.sh <rounds> <site> [<site> ...]
ALL="frame,iter,enc,gru32,stems,heads,refine,coarse"
rounds=$1; shift
run() { TCS_MI355_FORK_SITES=$1 python bench.py --steps 10 --warmup 3 --quick 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['step_ms_min_median_max'])"; }
for r in $(seq 1 $rounds); do
  echo "round $r all: $(run all)"
  for s in "$@"; do
    sites=$(echo $ALL | tr ',' '\n' | grep -v -x -E "$(echo $s | tr '+' '|')" | paste -sd, -)    # "a+b" removes both
    echo "round $r minus $s: $(run $sites)"
  done
done
