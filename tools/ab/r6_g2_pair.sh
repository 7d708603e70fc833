#!/bin/bash
# round 6: the G2 accumulation on lane pairs (zl_fq2pair.h, k_msm_accumulate_pair: two waves per SIMD) against the one-lane kernel (ZL_TUNE_G2_PAIR=0)
mkdir -p gpurun_out
out=gpurun_out/r06_g2_pair_ab.log
: > $out
timeout 900 python -m pytest tests/test_gpu_msm_g2.py tests/test_groth16.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2; do
for pair in 0 1; do
  echo "== ZL_TUNE_G2_PAIR=$pair" >> $out
  ZL_TUNE_G2_PAIR=$pair python tools/msm_sweep.py --g2 16 18 20 2>&1 | grep "2^" >> $out
  ZL_TUNE_G2_PAIR=$pair CURVE=bn254 python tools/msm_sweep.py --g2 16 20 2>&1 | grep "2^" >> $out
  ZL_TUNE_G2_PAIR=$pair ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_TUNE_G2_PAIR=$pair CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
done
cat $out
