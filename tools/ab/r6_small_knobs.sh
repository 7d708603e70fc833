#!/bin/bash
# round 6: small / mid MSMs (2^12 .. 2^18, both curves) after the prefix_small rewrite: four-lane thresholds and chunk length revisited
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_small_knobs.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_gpu_msm_g2.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $L
run() {  # name, env...
  name=$1; shift
  for curve in bls12_381 bn254; do
    echo "== $name $curve" >> $L
    env "$@" CURVE=$curve python tools/msm_sweep.py 12 14 16 18 2>&1 | grep -v "amdgpu.ids" >> $L
  done
}
for rep in 1 2; do
run default X=1
run noglv ZL_NO_GLV=1
run qacc16k ZL_TUNE_QUAD_ACC_CHUNKS=16384
run qacc32k ZL_TUNE_QUAD_ACC_CHUNKS=32768
run qlanes32k ZL_TUNE_QUAD_LANES=32768
run qlanes16k ZL_TUNE_QUAD_LANES=16384
run chunk16 ZL_TUNE_CHUNK=16
run chunk32 ZL_TUNE_CHUNK=32
done
cat $L
