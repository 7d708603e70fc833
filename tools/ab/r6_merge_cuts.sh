#!/bin/bash
# round 6: merge of cut buckets with one lane per chunk boundary (ZL_TUNE_MERGE_CUTS=1, product) against one lane per bucket
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_merge_cuts_ab.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_gpu_msm_g2.py tests/test_groth16.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
for rep in 1 2; do
for v in 0 1; do
  echo "== ZL_TUNE_MERGE_CUTS=$v" >> $L
  ZL_TUNE_MERGE_CUTS=$v BATCH=6 python tools/msm_sweep.py 20 22 24 2>&1 | grep "2^" >> $L
  ZL_TUNE_MERGE_CUTS=$v python tools/msm_sweep.py --g2 20 2>&1 | grep "2^" >> $L
  ZL_TUNE_MERGE_CUTS=$v CURVE=bn254 BATCH=6 python tools/msm_sweep.py 24 2>&1 | grep "2^" >> $L
done
done
cat $L
