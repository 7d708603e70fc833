#!/bin/bash
# A/B of two builds of the library on ONE box: ZL_BACKEND_LIB=tools/libzl_old.so against the in-tree build, interleaved
#   gpurun --timeout 900 -- 'bash tools/ab/r4_ab_lib.sh'
O=gpurun_out/ab; mkdir -p $O; : > $O/ab.log
for rep in 1 2; do
  for lib in tools/libzl_old.so openzl_amd/libzl_backend.so; do
    echo "== $lib" >> $O/ab.log
    ZL_BACKEND_LIB=$PWD/$lib python tools/msm_one.py 24 0 -1 3 2>&1 | grep "2^" >> $O/ab.log
    ZL_BACKEND_LIB=$PWD/$lib BATCH=6 python tools/msm_sweep.py 20 24 2>&1 | grep "2^" >> $O/ab.log
    ZL_BACKEND_LIB=$PWD/$lib python tools/ntt_one.py 24 6 2>&1 | tail -2 >> $O/ab.log
  done
done
cat $O/ab.log
