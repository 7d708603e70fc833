#!/bin/bash
# round 4: sort sharing between the a_query / b_g1_query MSMs of a proof (same scalars, same plan): on / off, large and small circuits
mkdir -p gpurun_out
out=gpurun_out/r04_g16_share_ab.log
: > $out
for mode in share noshare share noshare; do
  if [ $mode = noshare ]; then export ZL_NO_SORT_SHARE=1; else unset ZL_NO_SORT_SHARE; fi
  echo "== 958465 constraints, $mode" >> $out
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
for mode in share noshare; do
  if [ $mode = noshare ]; then export ZL_NO_SORT_SHARE=1; else unset ZL_NO_SORT_SHARE; fi
  echo "== small circuits, $mode" >> $out
  ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth16 >> $out
done
unset ZL_NO_SORT_SHARE
echo "== host trace of the large proof" >> $out
ZL_HOST_TRACE=1 ITERS=3 python tools/g16_one.py 4096 2>&1 | tail -80 >> $out
cat $out
