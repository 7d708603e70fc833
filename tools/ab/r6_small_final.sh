#!/bin/bash
# round 6: small / mid MSMs on the final defaults (prefix_small rewritten, chunk rule for half-scalars, four-lane accumulation up to 24 576 chunks, BN254 split outside 2^13 .. 2^17)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_small_final.log; : > $L
timeout 2300 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $L
for rep in 1 2; do
for curve in bls12_381 bn254; do
  echo "== $curve" >> $L
  CURVE=$curve BATCH=6 python tools/msm_sweep.py 10 12 13 14 15 16 17 18 19 20 2>&1 | grep -v "amdgpu.ids" >> $L
done
for k in 1 8 64; do ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
for k in 1 8 64; do CURVE=bn254 ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L; done
done
python tools/msm_sweep.py --g2 12 14 16 18 2>&1 | grep -v "amdgpu.ids" >> $L
cat $L
