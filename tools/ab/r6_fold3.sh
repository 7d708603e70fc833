#!/bin/bash
# round 6: fold threshold per curve
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_fold3_ab.log; : > $L
for rep in 1 2; do
for v in 0 20; do
  echo "== ZL_TUNE_G16_FOLD_LOG_N=$v" >> $L
  for k in 8 32 64; do
  ZL_TUNE_G16_FOLD_LOG_N=$v CURVE=bn254 ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
  done
  for k in 32 128 256; do
  ZL_TUNE_G16_FOLD_LOG_N=$v ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
done
done
cat $L
