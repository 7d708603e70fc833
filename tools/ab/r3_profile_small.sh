#!/bin/bash
# Timelines (per-launch start / duration / gap) of the small and mid-size calls: 2^16 / 2^20 MSM, Groth16 k = 1 and k = 64.
#   gpurun --timeout 900 -- 'bash tools/ab/r3_profile_small.sh <tag>'
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3small}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
ZL_HOST_TRACE=1 python tools/small_lat.py > $O/small_lat.log 2>&1
for ln in 16 20; do
  rocprofv3 --kernel-trace --stats -d $O/p$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/msm_one_$ln.log 2>&1
  python tools/timeline.py $(dbof $O/p$ln) 150 > $O/timeline_msm_2_$ln.txt 2>&1
  python tools/prof_summary.py $(dbof $O/p$ln) > $O/kernel_stats_msm_2_$ln.txt 2>&1
done
for k in 1 64; do
  rocprofv3 --kernel-trace --stats -d $O/pg$k -o t -- python tools/g16_one.py $k > $O/g16_one_$k.log 2>&1
  python tools/timeline.py $(dbof $O/pg$k) 150 > $O/timeline_g16_k$k.txt 2>&1
done
rm -rf $O/p16 $O/p20 $O/pg1 $O/pg64
ls -la $O
