#!/bin/bash
# round 6: the window tables inside mid-size proofs (four G1 MSMs + one G2 MSM of 2^16 .. 2^18 points side by side)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_tables_mid_ab.log; : > $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_HALF_TABLE=$v" >> $L
  for k in 64 128 256 512 1024; do ZL_TUNE_HALF_TABLE=$v ITERS=20 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  for k in 64 256 1024; do ZL_TUNE_HALF_TABLE=$v CURVE=bn254 ITERS=20 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L; done
done
done
cat $L
