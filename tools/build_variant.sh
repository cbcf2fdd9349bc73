#!/bin/bash
# Build a VARIANT of the library beside the production one for same-box A/B runs (bench.py picks it up through TCS_MI355_LIB):
#   tools/build_variant.sh <name> <extra hipcc flags...>   ->  temporally-consistent-stereo-matching_amd/lib/libtcs_mi355_<name>.so
# Only the translation units named in VARIANT_SOURCES (default: all) are recompiled with the extra flags; objects go to lib/variant_<name>/.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/temporally-consistent-stereo-matching_amd
out=$pkg/lib/variant_$name; mkdir -p $out
flags="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form"
srcs="tcs_corr tcs_warp tcs_stencil tcs_conv tcs_conv_f16 tcs_conv_s16 tcs_s16_ops"
objs=""
for s in $srcs; do
  if [[ " ${VARIANT_SOURCES:-$srcs} " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc $flags "$@" -c $pkg/csrc/$s.hip -o $out/$s.o &
    objs="$objs $out/$s.o"
  else
    objs="$objs $pkg/lib/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $pkg/lib/libtcs_mi355_$name.so $objs
echo $pkg/lib/libtcs_mi355_$name.so
