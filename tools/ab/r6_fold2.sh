#!/bin/bash
# round 6: (a) where the folded C query stops paying (k = 64 .. 1024 hashes: domains 2^14 .. 2^18), (b) window tables (one bucket set, no host Horner) for the small proof's three MSMs
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_fold2_ab.log; : > $L
timeout 900 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $L
ZL_TUNE_G16_FOLD_TABLE_C=16 timeout 900 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $L
for rep in 1 2; do
for v in 0 20; do
  echo "== ZL_TUNE_G16_FOLD_LOG_N=$v" >> $L
  for k in 64 128 256 1024; do
  ZL_TUNE_G16_FOLD_LOG_N=$v ITERS=24 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
  ZL_TUNE_G16_FOLD_LOG_N=$v CURVE=bn254 ITERS=24 python tools/g16_one.py 128 2>&1 | grep "prove k=" >> $L
done
for t in 0 16; do
  echo "== ZL_TUNE_G16_FOLD_TABLE_C=$t (fold up to 2^16)" >> $L
  for k in 1 8 64; do
  ZL_TUNE_G16_FOLD_LOG_N=16 ZL_TUNE_G16_FOLD_TABLE_C=$t ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
done
done
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 > $O/r06_fold2_trace_k1.log 2>&1
ZL_TUNE_G16_FOLD_TABLE_C=16 ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 > $O/r06_fold2_trace_k1_tab.log 2>&1
cat $L
