#!/bin/bash
# round 6: the witness map's sparse matrix-vector products with eight lanes per row (ZL_TUNE_SPMV8=1) against one lane per row
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_spmv8_ab.log; : > $L
timeout 900 python -m pytest tests/test_groth16.py tests/test_golden_vectors.py tests/test_gpu_lanes.py tests/test_gpu_key_wire.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_SPMV8=$v" >> $L
  for k in 1 8 64 256; do ZL_TUNE_SPMV8=$v ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  ZL_TUNE_SPMV8=$v ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  ZL_TUNE_SPMV8=$v CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $L
done
done
cat $L
