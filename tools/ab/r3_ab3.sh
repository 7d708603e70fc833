#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for q in 4 8; do for m in 0 1 2; do
  echo "== GPU_MAX_HW_QUEUES=$q ZL_TUNE_LANE_PRIO=$m"
  GPU_MAX_HW_QUEUES=$q ZL_TUNE_LANE_PRIO=$m ITERS=40 python tools/small_lat.py g16 2>&1 | grep "^Groth"
done; done
