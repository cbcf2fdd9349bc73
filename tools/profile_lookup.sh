#!/bin/bash
# The corr lookup under rocprofv3 at the three sizes SURVEY.md section 7 names (one 640x480 sequence, four sequences per launch, KITTI
# 375x1242) + the two PMC passes of the headline size, folded into <out>/lookup_pmc.json (tools/make_lookup_pmc_json.py).
# usage (inside a gpurun command): bash tools/profile_lookup.sh <dir under gpurun_out>
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$1
mkdir -p $out
# --quick: timed region only.  (With the stamped roofline pass behind it — graphs dropped and re-captured — the four-sequence run segfaults inside
# hipGraphLaunch UNDER rocprofv3, twice out of twice; without the profiler the same sequence is what bench.py's batched leg does every run.)
common="--warmup 3 --quick"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/p1 -- python3 $root/bench.py --steps 8 $common > $out/bench_1seq.json 2> $out/bench_1seq.err
rocprofv3 --kernel-trace --output-format csv -d $out/p4 -- python3 $root/bench.py --steps 4 --seqs-per-gpu 4 $common > $out/bench_4seq.json 2> $out/bench_4seq.err
rocprofv3 --kernel-trace --output-format csv -d $out/pk -- python3 $root/bench.py --steps 6 --size 375x1242 $common > $out/bench_kitti.json 2> $out/bench_kitti.err
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex k_corr_lookup --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --batched-leg 0 --drop-in-steps 0 --kitti-steps 0 --eager > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex k_corr_lookup --output-format csv -d $out/write -- python3 $root/bench.py --steps 2 --warmup 2 --no-cpu-baseline --batched-leg 0 --drop-in-steps 0 --kitti-steps 0 --eager > /dev/null 2> $out/pmc_write.err
cd $root
for p in p1 p4 pk; do
  trace=$(find $out/$p -name "*kernel_trace.csv" | head -1)
  [ -n "$trace" ] || { echo "no trace for $p"; tail -3 $out/bench_*.err; exit 1; }
  python tools/fold_kernel_trace.py $trace $out/fold_$p.csv > /dev/null
done
python tools/make_lookup_pmc_json.py --fetch $out/fetch --write $out/write --row one_sequence_640x480:$out/fold_p1.csv:19200 \
    --row four_sequences_640x480:$out/fold_p4.csv:76800 --row kitti_375x1242:$out/fold_pk.csv:29952 --out $out/lookup_pmc.json > /dev/null
rm -rf $out/p1 $out/p4 $out/pk $out/fetch $out/write
echo "lookup profiles folded into $out/lookup_pmc.json"
