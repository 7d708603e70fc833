#!/bin/bash
# round 6: the witness map's a / b / c transforms three at a time (ZL_TUNE_NTT_BATCH=1: one launch per pass for the three vectors) against one after the other (0)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_ntt_batch_ab.log; : > $L
timeout 900 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py tests/test_gpu_multi.py tests/test_gpu_ntt.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 >> $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_NTT_BATCH=$v" >> $L
  for k in 1 8 64 256 1024; do ZL_TUNE_NTT_BATCH=$v ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  ZL_TUNE_NTT_BATCH=$v ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  for k in 1 64; do ZL_TUNE_NTT_BATCH=$v CURVE=bn254 ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L; done
  ZL_TUNE_NTT_BATCH=$v CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
done
done
cat $L
