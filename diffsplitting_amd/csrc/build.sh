#!/bin/bash
# Builds libdsx.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libdsx.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DSX_EXTRA_FLAGS"
mkdir -p _obj
SRCS="dsx_conv.hip dsx_ops.hip dsx_attn.hip"
# a changed flag set rebuilds everything
if [ "$(cat _obj/.flags 2>/dev/null)" != "$FLAGS" ]; then rm -f _obj/*.o; echo "$FLAGS" > _obj/.flags; fi
pids=()
for f in $SRCS; do
  newer_inc=0
  for inc in *.inc; do [ "$inc" -nt _obj/$f.o ] && newer_inc=1; done     # included bodies (dsx_conv_ws_item.inc)
  if [ ! -f _obj/$f.o ] || [ $f -nt _obj/$f.o ] || [ dsx_kernels.h -nt _obj/$f.o ] || [ $newer_inc = 1 ]; then
    rm -f _obj/$f.o                       # a failed compile must not leave a stale object to link
    hipcc $FLAGS -c $f -o _obj/$f.o &
    pids+=($!)
  fi
done
if [ ! -f _obj/rt.o ] || [ dsx_runtime.cpp -nt _obj/rt.o ] || [ dsx_kernels.h -nt _obj/rt.o ] || [ ../../include/dsx.h -nt _obj/rt.o ]; then
  rm -f _obj/rt.o
  hipcc $FLAGS -x hip -c dsx_runtime.cpp -o _obj/rt.o &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p || { echo "build.sh: a compile failed" >&2; exit 1; }; done
OBJS=""
for f in $SRCS; do OBJS="$OBJS _obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS _obj/rt.o
echo "built $(realpath $OUT)"
