#!/bin/bash
# Experiment builds: tools/build_variant.sh NAME FILE.hip [-DKNOB=...]  recompiles ONE translation unit with extra flags and links it
# with the other objects of the regular build into _variants/NAME/libsdpgpu.so (select with SDPGPU_LIB=...; *.so is git-ignored).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; file=$2; shift 2
mkdir -p "$R/_variants/$name"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wno-unused-function -fvisibility=hidden \
  "$@" -c -o "$R/_variants/$name/$file.o" "$R/stochastic-inventory_amd/csrc/$file"
objs=""
for f in "$R"/stochastic-inventory_amd/_build/*.hip.o; do
  if [ "$(basename "$f")" = "$file.o" ]; then objs="$objs $R/_variants/$name/$file.o"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/_variants/$name/libsdpgpu.so" $objs -lhiprtc
echo "$R/_variants/$name/libsdpgpu.so"
