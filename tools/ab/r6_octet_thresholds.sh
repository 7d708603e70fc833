#!/bin/bash
# round 6: lane-count thresholds of the eight-lane G2 forms (shared knobs with the G1 four-lane forms: G2 sizes and proofs only here)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_octet_thresholds.log; : > $L
for rep in 1 2; do
for ql in 16384 32768 65536 131072; do
for qa in 24576 49152 98304; do
  echo "== ZL_TUNE_QUAD_LANES=$ql ZL_TUNE_QUAD_ACC_CHUNKS=$qa" >> $L
  ZL_TUNE_QUAD_LANES=$ql ZL_TUNE_QUAD_ACC_CHUNKS=$qa python tools/msm_sweep.py --g2 12 14 16 18 2>&1 | grep "2^" | awk '{printf "%s %s wall %s dev %s acc %s | ", $1, $2, $4, $7, $10} END {print ""}' >> $L
  ZL_TUNE_QUAD_LANES=$ql ZL_TUNE_QUAD_ACC_CHUNKS=$qa ITERS=20 python tools/g16_one.py 64 2>&1 | grep "prove k=" >> $L
done
done
done
cat $L
