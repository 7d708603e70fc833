#!/bin/bash
# round 6: is the endomorphism split still a gain on BLS12-381 G1 at small sizes (measured in round 3, before the tails were shortened)?
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_bls_glv_ab.log; : > $L
for rep in 1 2 3; do
for v in 1 0; do
  if [ $v = 1 ]; then export ZL_NO_GLV=1; else unset ZL_NO_GLV; fi
  echo "== ZL_NO_GLV=$v" >> $L
  BATCH=6 python tools/msm_sweep.py 10 12 14 16 18 19 2>&1 | grep -v amdgpu.ids >> $L
  for k in 1 8 64; do
  ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
done
done
cat $L
