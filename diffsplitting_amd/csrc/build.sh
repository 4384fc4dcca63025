#!/bin/bash
# Builds libdsx.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libdsx.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DSX_EXTRA_FLAGS"
mkdir -p _obj
for f in dsx_conv.hip dsx_ops.hip; do
  if [ ! -f _obj/$f.o ] || [ $f -nt _obj/$f.o ] || [ dsx_kernels.h -nt _obj/$f.o ]; then
    hipcc $FLAGS -c $f -o _obj/$f.o &
  fi
done
if [ ! -f _obj/rt.o ] || [ dsx_runtime.cpp -nt _obj/rt.o ] || [ dsx_kernels.h -nt _obj/rt.o ] || [ ../../include/dsx.h -nt _obj/rt.o ]; then
  hipcc $FLAGS -x hip -c dsx_runtime.cpp -o _obj/rt.o &
fi
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT _obj/dsx_conv.hip.o _obj/dsx_ops.hip.o _obj/rt.o
echo "built $(realpath $OUT)"
