#!/bin/bash
# Diagnostic: build variants of the library with one phase of the f16x3 conv main loop removed
# (-DTCS_ABLATE_LOAD / _STORE / _MMA; outputs are garbage) and time the loop's layer shapes with each,
# to see which phase the wall time follows.  Build here (no GPU needed), run tools/bench_conv.py on the GPU box with
# TCS_MI355_LIB=<variant>.
set -e
cd "$(dirname "$0")/../temporally-consistent-stereo-matching_amd"
for v in LOAD STORE MMA "LOAD -DTCS_ABLATE_STORE"; do
  name=$(echo "$v" | tr -d ' ' | sed 's/-DTCS_ABLATE_/_/')
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -shared -DTCS_ABLATE_$v \
     csrc/tcs_corr.hip csrc/tcs_warp.hip csrc/tcs_stencil.hip csrc/tcs_conv.hip csrc/tcs_conv_f16.hip -o lib/libtcs_ablate_$name.so &
done
wait
ls -la lib/
