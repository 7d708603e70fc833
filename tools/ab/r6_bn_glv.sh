#!/bin/bash
# round 6: BN254 G1 with the endomorphism split (two-dimensional lattice decomposition) against the plain 254-bit windows (ZL_NO_GLV=1), and where it stops paying (ZL_TUNE_GLV_MAX_LOG)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_bn_glv_ab.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_groth16.py tests/test_gpu_lanes.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 >> $L
for rep in 1 2 3; do
for v in 1 0; do
  if [ $v = 1 ]; then export ZL_NO_GLV=1; else unset ZL_NO_GLV; fi
  echo "== ZL_NO_GLV=$v" >> $L
  CURVE=bn254 BATCH=6 python tools/msm_sweep.py 12 14 16 18 19 2>&1 | grep -v amdgpu.ids >> $L
  for k in 1 8 64; do
  CURVE=bn254 ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
  done
done
done
unset ZL_NO_GLV
echo "== ZL_TUNE_GLV_MAX_LOG=21 at 2^20, 2^21 (against plain)" >> $L
for rep in 1 2; do
ZL_TUNE_GLV_MAX_LOG=21 CURVE=bn254 BATCH=6 python tools/msm_sweep.py 20 21 2>&1 | grep -v amdgpu.ids | sed 's/^/glv  /' >> $L
CURVE=bn254 BATCH=6 python tools/msm_sweep.py 20 21 2>&1 | grep -v amdgpu.ids | sed 's/^/plain /' >> $L
done
cat $L
