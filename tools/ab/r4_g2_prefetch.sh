#!/bin/bash
# round 4: the G2 accumulation's next base through LDS (global_load_lds, in-tree build) against the direct gather (-DZL_NO_ACC_PREFETCH = tools/libzl_nopf.so)
mkdir -p gpurun_out
out=gpurun_out/r04_g2_prefetch_ab.log
: > $out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for lib in tools/libzl_nopf.so openzl_amd/libzl_backend.so; do
  echo "== $lib" >> $out
  ZL_BACKEND_LIB=$PWD/$lib python tools/msm_sweep.py --g2 12 16 20 2>&1 | grep "2^" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 python tools/msm_sweep.py --g2 16 20 2>&1 | grep "2^" | sed 's/^/bn254 /' >> $out
  ZL_BACKEND_LIB=$PWD/$lib ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $out
  ZL_BACKEND_LIB=$PWD/$lib ITERS=30 python tools/g16_one.py 64 2>&1 | grep "prove k=" >> $out
done
done
cat $out
