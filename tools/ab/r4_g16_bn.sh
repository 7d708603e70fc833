#!/bin/bash
# round 4: Groth16 over BN254 at the config-5 size (the reference's other curve), kernel table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g16bn; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
export CURVE=bn254
ITERS=10 python tools/g16_one.py 4096 2>&1 | grep -E "prove k=|synthesis" > $O/log.txt
ITERS=20 python tools/g16_one.py 64 2>&1 | grep -E "prove k=" >> $O/log.txt
rocprofv3 --kernel-trace --stats -d $O/p5 -o t -- python tools/g16_one.py > $O/g16_one.log 2>&1
python tools/prof_summary.py $(dbof $O/p5) > $O/kernel_stats_groth16_bn254.txt
python tools/timeline.py $(dbof $O/p5) 1500 1 150 > $O/g16_timeline_bn254.txt 2>&1
rm -rf $O/p5
cat $O/log.txt; head -30 $O/kernel_stats_groth16_bn254.txt | cut -c1-72,118-200
