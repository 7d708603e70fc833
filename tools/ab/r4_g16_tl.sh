#!/bin/bash
# timeline of the large proof (launches >= 150 us of the last proof) + kernel stats, under rocprofv3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g16tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
rocprofv3 --kernel-trace --stats -d $O/p5 -o t -- python tools/g16_one.py > $O/g16_one.log 2>&1
python tools/prof_summary.py $(dbof $O/p5) > $O/kernel_stats_groth16.txt
python tools/timeline.py $(dbof $O/p5) 1500 1 ${TL_MIN_US:-150} > $O/g16_timeline.txt 2>&1
rm -rf $O/p5
tail -3 $O/g16_one.log
