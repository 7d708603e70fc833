#!/bin/bash
# round 4: the large proof x hardware queues per process (GPU_MAX_HW_QUEUES) x earlier work in the process (PRE_LEGS)
mkdir -p gpurun_out
out=gpurun_out/r04_g16_hwq_ab.log
: > $out
for rep in 1 2; do
for pre in 0 1; do
for q in 4 6 8; do
  if [ $pre = 1 ]; then export PRE_LEGS=1; else unset PRE_LEGS; fi
  export GPU_MAX_HW_QUEUES=$q
  echo "== 958465 constraints, PRE_LEGS=$pre GPU_MAX_HW_QUEUES=$q" >> $out
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
done
done
unset PRE_LEGS
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  for k in 1 64; do
    echo "== k=$k GPU_MAX_HW_QUEUES=$q" >> $out
    python tools/g16_lat_dist.py $k 60 2>&1 | grep "after 3" >> $out
  done
  echo "== 2^24 batch, 2^20 GPU_MAX_HW_QUEUES=$q" >> $out
  BATCH=6 python tools/msm_sweep.py 20 24 2>&1 | grep "2^" >> $out
done
cat $out
