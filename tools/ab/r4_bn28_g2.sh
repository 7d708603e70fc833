#!/bin/bash
# round 4: BN254 G2 on the lazily reduced 28-bit Fq2 (in-tree: one wave per SIMD; libzl_bng2_w2: two waves + 244 B scratch) against the 32-bit fields (libzl_bn32)
mkdir -p gpurun_out
out=gpurun_out/r04_bn254_g2_field28_ab.log
: > $out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for lib in tools/libzl_bn32.so openzl_amd/libzl_backend.so tools/libzl_bng2_w2.so; do
  echo "== $lib" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 python tools/msm_sweep.py --g2 12 16 20 2>&1 | grep "2^" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 ITERS=30 python tools/g16_one.py 64 2>&1 | grep "prove k=" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 ITERS=30 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $out
done
done
ZL_BACKEND_LIB=$PWD/tools/libzl_bng2_w2.so python -m pytest tests/test_groth16.py tests/test_gpu_msm_g2.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" >> $out
cat $out
