#!/bin/bash
# the small-lane classes measured in the stream population bench.py has when it reaches its Groth16 leg (PRE_LEGS=1: a pipelined batch and a host-scalar MSM first)
for rep in 1 2; do
for cfg in "0 4" "2 3" "2 2" "1 3"; do
  set -- $cfg
  for k in 1 64; do
    echo "PRE_LEGS SMALL_LANE_PRIO=$1 LANES=$2: $(PRE_LEGS=1 ZL_TUNE_SMALL_LANE_PRIO=$1 ZL_TUNE_SMALL_LANES=$2 ZL_TUNE_SIDE_LANES=4 ITERS=40 python tools/g16_one.py $k 2>&1 | tail -1)"
  done
done
done
for p in 0 2; do echo "PRE_LEGS SMALL_LANE_PRIO=$p: $(PRE_LEGS=1 ZL_TUNE_SMALL_LANE_PRIO=$p ITERS=9 python tools/g16_one.py 4096 2>&1 | tail -1)"; done
