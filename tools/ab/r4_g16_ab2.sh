#!/bin/bash
# round 4: event pool (ctx-owned events instead of create / destroy per pipeline) and, TIMING ONLY, the a / b1 jobs forced to share one sort although their
# infinity sets differ (ZL_EXPERIMENT_FORCE_SHARE: the result is wrong, the schedule is what a correct shared sort would give at best)
mkdir -p gpurun_out
out=gpurun_out/r04_g16_eventpool_ab.log
: > $out
for mode in pool forceshare pool forceshare; do
  if [ $mode = forceshare ]; then export ZL_EXPERIMENT_FORCE_SHARE=1; else unset ZL_EXPERIMENT_FORCE_SHARE; fi
  echo "== 958465 constraints, $mode" >> $out
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
unset ZL_EXPERIMENT_FORCE_SHARE
echo "== small circuits" >> $out
ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth16 >> $out
echo "== host trace of the large proof" >> $out
ZL_HOST_TRACE=1 ITERS=3 python tools/g16_one.py 4096 2>&1 | tail -28 >> $out
cat $out
