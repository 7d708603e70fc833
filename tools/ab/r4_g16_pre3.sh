#!/bin/bash
# round 4: small and large proofs x PRE_LEGS x priority class of the copy stream (ZL_TUNE_COPY_PRIO=1: with the sort / tail streams; 0: default class)
mkdir -p gpurun_out
out=gpurun_out/r04_g16_pre3.log
: > $out
for rep in 1 2; do
for cp in 1 0; do
for pre in 0 1; do
  if [ $pre = 1 ]; then export PRE_LEGS=1; else unset PRE_LEGS; fi
  export ZL_TUNE_COPY_PRIO=$cp
  echo "== COPY_PRIO=$cp PRE_LEGS=$pre" >> $out
  ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  python tools/g16_lat_dist.py 1 50 2>&1 | grep "after 3" >> $out
  python tools/g16_lat_dist.py 64 40 2>&1 | grep "after 3" >> $out
done
done
done
cat $out
