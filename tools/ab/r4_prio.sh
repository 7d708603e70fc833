#!/bin/bash
# round 4: all of the library's streams in ONE priority class (ZL_TUNE_STREAM_PRIO=0: at most four hardware queues for the process) against sort / tail / witness-map /
# copy streams in the high-priority class (default: two classes, up to eight queues)
mkdir -p gpurun_out
out=gpurun_out/r04_stream_prio_ab.log
: > $out
for rep in 1 2; do
for pr in 1 0; do
for pre in 0 1; do
  if [ $pre = 1 ]; then export PRE_LEGS=1; else unset PRE_LEGS; fi
  export ZL_TUNE_STREAM_PRIO=$pr
  echo "== STREAM_PRIO=$pr PRE_LEGS=$pre" >> $out
  ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  python tools/g16_lat_dist.py 1 50 2>&1 | grep "after 3" >> $out
  python tools/g16_lat_dist.py 64 40 2>&1 | grep "after 3" >> $out
done
unset PRE_LEGS
BATCH=6 python tools/msm_sweep.py 16 20 24 2>&1 | grep "BATCH" >> $out
done
done
cat $out
