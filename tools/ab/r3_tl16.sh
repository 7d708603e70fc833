#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-tl16}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for ln in ${2:-16}; do
rocprofv3 --kernel-trace --stats -d $O/p$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/msm_one_$ln.log 2>&1
python tools/timeline.py $(find $O/p$ln -name "*.db" | head -1) 150 | cut -c1-125
rm -rf $O/p$ln
done
