#!/bin/bash
# rocprofv3 kernel trace of a short bench.py run on the GPU box, folded into the files profiles/ keeps:
#   <out>/iteration_timeline.txt (tools/iter_timeline.py), <out>/kernel_trace_fold.csv (tools/fold_kernel_trace.py), <out>/kernel_stats.csv,
#   <out>/frame_phases.txt (tools/frame_phases.py: head / loop / tail of the last frames)
# usage (inside a gpurun command): bash tools/profile_bench.sh <dir under gpurun_out> [extra bench.py arguments]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $root/bench.py --steps 8 --warmup 3 --no-cpu-baseline --batched-leg 0 --drop-in-steps 0 --kitti-steps 0 "$@" \
    > $out/bench_prof.json 2> $out/bench_prof.err
cd $root
trace=$(find $out/prof -name "*kernel_trace.csv" | head -1)
stats=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$trace" ] || { echo "no kernel trace produced"; tail -5 $out/bench_prof.err; exit 1; }
python tools/iter_timeline.py $trace > $out/iteration_timeline.txt
python tools/fold_kernel_trace.py $trace $out/kernel_trace_fold.csv
python tools/frame_phases.py $trace 8 > $out/frame_phases.txt
[ -n "$stats" ] && cp $stats $out/kernel_stats.csv
rm -rf $out/prof
echo "profile folded into $out"
