#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3prof2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
for ln in 20 24; do
  rocprofv3 --kernel-trace --stats -d $O/p$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/msm_one_$ln.log 2>&1
  python tools/timeline.py $(dbof $O/p$ln) 150 > $O/timeline_msm_2_$ln.txt 2>&1
done
rm -rf $O/p20 $O/p24
cat $O/timeline_msm_2_20.txt; cat $O/timeline_msm_2_24.txt
