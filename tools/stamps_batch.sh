export DSX_LIB_PATH=$PWD/diffsplitting_amd/csrc/_variants/libdsx_stamps.so
for op in ${STAMP_OPS:-1 15 22 28 21}; do
  echo "=== op $op"
  DSX_STAMP_OP=$op timeout -k 10 120 python tools/stamps_raw.py 2>&1 | grep -v amdgpu.ids || exit 1
done
