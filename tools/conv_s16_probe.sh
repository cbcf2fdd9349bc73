#!/bin/bash
# Diagnostic build of the library with a one-store LINEAR epilogue in k_conv_s16 (-DTCS_S16_PROBE_SLIM): how fast is the K loop of a tile
# configuration when nothing but the K loop decides the register allocation?  Container: `bash tools/conv_s16_probe.sh` builds
# lib/libtcs_probe.so next to the product library; GPU box: TCS_MI355_LIB=$PWD/.../lib/libtcs_probe.so python tools/bench_conv_s16.py ...
# (the "!!ERR" marks are expected: the probe's outputs are not the convolution's).
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/temporally-consistent-stereo-matching_amd
flags=$(python3 -c "import sys; sys.path.insert(0, '$pkg'); from tcs_mi355 import build as b; print(' '.join(b.FLAGS))")
hipcc $flags -DTCS_S16_PROBE_SLIM -c $pkg/csrc/tcs_conv_s16.hip -o $pkg/lib/tcs_conv_s16_probe.o
objs=$(ls $pkg/lib/tcs_*.o | grep -v -e ablate -e probe -e "tcs_conv_s16.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/libtcs_probe.so $objs $pkg/lib/tcs_conv_s16_probe.o
echo built $pkg/lib/libtcs_probe.so
