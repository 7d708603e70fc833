#!/bin/bash
# round 6: eight lanes per G2 group operation (quad x pair, ZL_TUNE_G2_OCTET=1) against the four-lane one-lane-per-half forms, in the launches that do not fill the machine
mkdir -p gpurun_out
out=gpurun_out/r06_g2_octet_ab.log
: > $out
timeout 900 python -m pytest tests/test_gpu_msm_g2.py tests/test_groth16.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for o in 0 1; do
  echo "== ZL_TUNE_G2_OCTET=$o" >> $out
  ZL_TUNE_G2_OCTET=$o python tools/msm_sweep.py --g2 8 12 16 18 2>&1 | grep "2^" >> $out
  ZL_TUNE_G2_OCTET=$o CURVE=bn254 python tools/msm_sweep.py --g2 8 12 16 2>&1 | grep "2^" >> $out
  ZL_TUNE_G2_OCTET=$o ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $out
  ZL_TUNE_G2_OCTET=$o ITERS=40 python tools/g16_one.py 64 2>&1 | grep "prove k=" >> $out
  ZL_TUNE_G2_OCTET=$o ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_TUNE_G2_OCTET=$o CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $out
  ZL_TUNE_G2_OCTET=$o CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
done
cat $out
