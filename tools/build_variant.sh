#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..."  -> diffsplitting_amd/csrc/_variants/libdsx_NAME.so
# (diagnostic builds for timing experiments: select with DSX_LIB_PATH; the .so travels to the GPU box)
set -e
cd "$(dirname "$0")/../diffsplitting_amd/csrc"
NAME=$1; EXTRA=$2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA"
O=_obj_$NAME; mkdir -p $O _variants
pids=()
for f in dsx_conv.hip dsx_ops.hip dsx_attn.hip; do hipcc $FLAGS -c $f -o $O/$f.o & pids+=($!); done
hipcc $FLAGS -x hip -c dsx_runtime.cpp -o $O/rt.o & pids+=($!)
for p in "${pids[@]}"; do wait $p || { echo "variant $NAME: compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/libdsx_$NAME.so $O/dsx_conv.hip.o $O/dsx_ops.hip.o $O/dsx_attn.hip.o $O/rt.o
rm -rf $O
echo "built _variants/libdsx_$NAME.so"
