#!/bin/bash
# round 6: small / mid MSMs after the chunk rule for half-scalars: four-lane accumulation threshold revisited, both curves, and the small proofs as a regression check
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_small_knobs2.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $L
run() {  # name, env...
  name=$1; shift
  for curve in bls12_381 bn254; do
    echo "== $name $curve" >> $L
    env "$@" CURVE=$curve python tools/msm_sweep.py 10 12 13 14 15 16 17 18 19 2>&1 | grep -v "amdgpu.ids" >> $L
  done
}
for rep in 1 2; do
run default X=1
run noglv ZL_NO_GLV=1
run qacc16k ZL_TUNE_QUAD_ACC_CHUNKS=16384
run qacc24k ZL_TUNE_QUAD_ACC_CHUNKS=24576
run qlanes128k ZL_TUNE_QUAD_LANES=131072
done
for rep in 1 2; do
for v in 49152 16384; do
  echo "== proofs ZL_TUNE_QUAD_ACC_CHUNKS=$v" >> $L
  for k in 1 8 64; do ZL_TUNE_QUAD_ACC_CHUNKS=$v ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  for k in 1 64; do ZL_TUNE_QUAD_ACC_CHUNKS=$v CURVE=bn254 ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L; done
done
done
