#!/bin/bash
# round 4: BN254 base field on 10 x 28-bit lazily reduced limbs (in-tree build) against the 8 x 32-bit carry-chain field (-DZL_BN_FIELD32: tools/libzl_bn32.so)
mkdir -p gpurun_out
out=gpurun_out/r04_bn254_field28_ab.log
: > $out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for lib in tools/libzl_bn32.so openzl_amd/libzl_backend.so; do
  echo "== $lib" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 BATCH=6 python tools/msm_sweep.py 16 18 20 22 24 2>&1 | grep "2^" >> $out
done
done
cat $out
