#!/bin/bash
# round 4: phase-major issue of the four G1 jobs + the G2 accumulation held back until their sorts finished (large circuits): on / off
mkdir -p gpurun_out
out=gpurun_out/r04_g16_gate_ab.log
: > $out
python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_golden_vectors.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" >> $out
for mode in gate nogate gate nogate; do
  if [ $mode = nogate ]; then export ZL_TUNE_G2_GATE_MIN_LOG=40 ZL_TUNE_PHASED_MIN_LOG=40; else unset ZL_TUNE_G2_GATE_MIN_LOG ZL_TUNE_PHASED_MIN_LOG; fi
  echo "== 958465 constraints, $mode" >> $out
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
export ZL_TUNE_G2_GATE_MIN_LOG=40; unset ZL_TUNE_PHASED_MIN_LOG
echo "== 958465 constraints, phased only" >> $out
ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
unset ZL_TUNE_G2_GATE_MIN_LOG ZL_TUNE_PHASED_MIN_LOG
for k in 64 512; do
for mode in gate nogate; do
  if [ $mode = nogate ]; then export ZL_TUNE_G2_GATE_MIN_LOG=40 ZL_TUNE_PHASED_MIN_LOG=40; else export ZL_TUNE_G2_GATE_MIN_LOG=10 ZL_TUNE_PHASED_MIN_LOG=10; fi
  echo "== k=$k, $mode (thresholds forced)" >> $out
  ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $out
done
done
unset ZL_TUNE_G2_GATE_MIN_LOG ZL_TUNE_PHASED_MIN_LOG
cat $out
TL_MIN_US=150 bash tools/ab/r4_g16_tl.sh
