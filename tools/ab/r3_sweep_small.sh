#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3sweep}
mkdir -p $O
cd $R
for CH in 8 16 32 64; do for SEG in 1 2 4 8; do
  echo "== CHUNK $CH SEG $SEG" >> $O/sweep.log
  SIZES=14,16,18 ZL_TUNE_CHUNK=$CH ZL_TUNE_SEG=$SEG python tools/small_lat.py msm 2>&1 | grep "^MSM" >> $O/sweep.log
done; done
echo "== default (no tuning)" >> $O/sweep.log
SIZES=14,16,18,20 python tools/small_lat.py msm 2>&1 | grep "^MSM" >> $O/sweep.log
for SEG in 1 2 4; do echo "== 2^20 SEG $SEG" >> $O/sweep.log; SIZES=20,22 ZL_TUNE_SEG=$SEG python tools/small_lat.py msm 2>&1 | grep "^MSM" >> $O/sweep.log; done
cat $O/sweep.log
