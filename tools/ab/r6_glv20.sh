#!/bin/bash
# round 6: the endomorphism split up to 2^20 points (ZL_TUNE_GLV_MAX_LOG=20; BLS12-381 at c = 16 from the table) against 2^19 (default so far): config 2 and the proofs whose MSMs are that size
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_glv20b_ab.log; : > $L
for rep in 1 2 3; do
for v in 19 20; do
  echo "== ZL_TUNE_GLV_MAX_LOG=$v" >> $L
  ZL_TUNE_GLV_MAX_LOG=$v BATCH=6 python tools/msm_sweep.py 20 2>&1 | grep -v amdgpu.ids >> $L
  ZL_TUNE_GLV_MAX_LOG=$v CURVE=bn254 BATCH=6 python tools/msm_sweep.py 20 2>&1 | grep -v amdgpu.ids | sed 's/^/bn254 /' >> $L
  ZL_TUNE_GLV_MAX_LOG=$v ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  ZL_TUNE_GLV_MAX_LOG=$v ITERS=10 python tools/g16_one.py 3000 2>&1 | grep "prove k=" >> $L
  ZL_TUNE_GLV_MAX_LOG=$v CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
done
done
ZL_TUNE_GLV_MAX_LOG=20 timeout 600 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 >> $L
cat $L
