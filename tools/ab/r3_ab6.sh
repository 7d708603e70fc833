#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
for l in 4 3 2 1; do
  echo "== ZL_TUNE_SIDE_LANES=$l (fresh process)"
  ZL_TUNE_SIDE_LANES=$l ITERS=40 timeout 300 python tools/small_lat.py g16 2>&1 | grep "^Groth"
done
for l in 4 2; do
  echo "== ZL_TUNE_SIDE_LANES=$l after a pipelined 2^20 batch in the same process (all streams of the ctx exist)"
  ZL_TUNE_SIDE_LANES=$l ITERS=40 SIZES=22 timeout 300 python tools/small_lat.py msm g16 2>&1 | grep "^Groth"
done
