#!/bin/bash
# A/B of end-to-end frame time inside ONE gpurun call (boxes differ by several percent): interleaved rounds of
# `bench.py --steps 20` per variant.  A variant is a value of TCS_MI355_X ("-" = the default build), or, when it starts with "--",
# extra bench.py arguments (e.g. --no-prefetch); "tokens@--args" combines both.
# usage: tools/ab_bench.sh <rounds> <variant> [<variant> ...]
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    x=${v%%@*}; args=""
    [[ "$v" == *@* ]] && args=${v#*@}
    [[ "$x" == --* ]] && { args="$x $args"; x=""; }
    [ "$x" = "-" ] && x=""
    ms=$(TCS_MI355_X=$x python bench.py --steps 20 --warmup 3 --quick $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'per-step min/median/max', d['step_ms_min_median_max'])")
    echo "round $r variant [$v] ms_per_step $ms"
  done
done
