#!/bin/bash
# round 6: the window picker's per-bucket cost 5.2 -> 4.5: pipelined sweeps of every window at 2^21 .. 2^23 on both curves (the first line of a size is the picker's own choice)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_window_sweep_2_21_23.log; : > $L
for rep in 1 2; do
for cv in bls12_381 bn254; do
  echo "== $cv (picker at 4.5 first, then forced windows)" >> $L
  CURVE=$cv BATCH=6 CS=0,16,17,18,19,20 python tools/msm_sweep.py 21 22 23 2>&1 | grep -v amdgpu.ids >> $L
done
done
for cv in bls12_381 bn254; do
  echo "== $cv picker, cost 5.2 | 4.5, all sizes" >> $L
  ZL_TUNE_BUCKET_COST_X10=52 CURVE=$cv BATCH=6 python tools/msm_sweep.py 10 12 14 16 18 19 20 21 22 23 24 2>&1 | grep -v amdgpu.ids | sed 's/^/5.2  /' >> $L
  CURVE=$cv BATCH=6 python tools/msm_sweep.py 10 12 14 16 18 19 20 21 22 23 24 2>&1 | grep -v amdgpu.ids | sed 's/^/4.5  /' >> $L
done
timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 >> $L
