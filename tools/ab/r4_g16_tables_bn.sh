#!/bin/bash
# round 4: window widths of the proving key's tables for BN254 (fields on 28-bit limbs), large proof
mkdir -p gpurun_out
out=gpurun_out/r04_g16_tables_bn254.log
: > $out
export CURVE=bn254
for c1 in 20 19 21; do
for c2 in 16 17 18; do
  export ZL_TUNE_G1_TABLE_C=$c1 ZL_TUNE_G2_TABLE_C=$c2
  echo "== BN254 G1 c=$c1 G2 c=$c2" >> $out
  ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
done
export ZL_TUNE_G1_TABLE_C=20 ZL_TUNE_G2_TABLE_C=16
echo "== BN254 G1 c=20 G2 c=16 (again)" >> $out
ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
cat $out
