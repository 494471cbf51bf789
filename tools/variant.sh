#!/bin/bash
# An A/B build of one translation unit: tools/variant.sh <name> <file.hip> <extra hipcc flags...>
#   -> image_processing_amd/lib/libipx_<name>.so (the other objects are those of the current build); use with IPX_LIB=...
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
L=image_processing_amd/lib
extra=""
[ "$src" = flow_iter.hip ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-result $extra "$@" \
    -c image_processing_amd/csrc/$src -o $L/${src%.hip}_$name.o
objs=""
for o in api detect temporal ccl flow flow_iter chunk group; do
  if [ "$o.hip" = "$src" ]; then objs="$objs $L/${o}_$name.o"; else objs="$objs $L/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libipx_$name.so $objs -ldl -lpthread
echo $L/libipx_$name.so
