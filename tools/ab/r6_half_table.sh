#!/bin/bash
# round 6: measured window table for split scalars on G1 (ZL_TUNE_HALF_TABLE=0: the cost model) + a window sweep of the G2 MSM at small sizes
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_half_table_ab.log; : > $L
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -3 >> $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_HALF_TABLE=$v" >> $L
  ZL_TUNE_HALF_TABLE=$v BATCH=6 python tools/msm_sweep.py 11 12 13 14 15 16 17 18 19 2>&1 | grep -v amdgpu.ids >> $L
  ZL_TUNE_HALF_TABLE=$v CURVE=bn254 BATCH=6 python tools/msm_sweep.py 11 12 17 18 19 2>&1 | grep -v amdgpu.ids | sed 's/^/bn254 /' >> $L
  for k in 8 64; do ZL_TUNE_HALF_TABLE=$v ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
done
done
echo "== G2 window sweep" >> $L
for rep in 1 2; do
CS=0,8,9,10,11,12,13,14,15,16 python tools/msm_sweep.py --g2 10 12 13 14 15 16 17 18 2>&1 | grep -v amdgpu.ids | sed 's/^/g2 /' >> $L
done
