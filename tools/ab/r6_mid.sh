#!/bin/bash
# round 6 mid-round check: GPU suite, bench line, G2 kernel table, proof timeline, c = 19 / 20 kernel tables at 2^24
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mid
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/gpu_tests.log
python bench.py > $O/r06_bench_mid.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/p6 -o t -- python tools/msm_sweep.py --g2 20 > $O/g2_sweep.log 2>&1
python tools/prof_summary.py $(dbof $O/p6) > $O/r06_kernel_stats_g2_2_20.txt
rocprofv3 --kernel-trace --stats -d $O/p5 -o t -- python tools/g16_one.py > $O/g16_one.log 2>&1
python tools/prof_summary.py $(dbof $O/p5) > $O/r06_kernel_stats_groth16.txt
python tools/timeline.py $(dbof $O/p5) 1500 1 200 > $O/r06_g16_timeline.txt 2>&1
for c in 19 20; do
  rocprofv3 --kernel-trace --stats -d $O/pc$c -o t -- python tools/msm_one.py 24 $c -1 3 > $O/msm_c$c.log 2>&1
  python tools/prof_summary.py $(dbof $O/pc$c) reduce_tree > $O/r06_kernel_stats_msm_2_24_c$c.txt
done
BATCH=6 CS=19,20 python tools/msm_sweep.py 24 > $O/r06_msm_sweep_c19_c20.log 2>&1
rm -rf $O/p5 $O/p6 $O/pc19 $O/pc20
cat $O/gpu_tests.log; python tools/bench_digest.py $O/r06_bench_mid.json 2>/dev/null | head -30
