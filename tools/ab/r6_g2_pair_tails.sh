#!/bin/bash
# round 6: the G2 merge / level-0 / tree kernels on lane pairs (ZL_TUNE_G2_PAIR_TAILS=1) against the one-lane forms
mkdir -p gpurun_out
out=gpurun_out/r06_g2_pair_tails_ab.log
: > $out
timeout 900 python -m pytest tests/test_gpu_msm_g2.py tests/test_groth16.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for pair in 0 1; do
  echo "== ZL_TUNE_G2_PAIR_TAILS=$pair" >> $out
  ZL_TUNE_G2_PAIR_TAILS=$pair python tools/msm_sweep.py --g2 18 20 22 2>&1 | grep "2^" >> $out
  ZL_TUNE_G2_PAIR_TAILS=$pair CURVE=bn254 python tools/msm_sweep.py --g2 18 20 2>&1 | grep "2^" >> $out
done
done
cat $out
