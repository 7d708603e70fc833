#!/bin/bash
# the headline leg alone under rocprofv3: big_avg_us of k_msm_accumulate in the table == roofline.kernel_ms of the JSON line of the same command
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p7 -o t -- python bench.py --no-skew --fixed-key -1 --no-ntt --groth16-k 0 --no-cpu --no-configs > $O/r03_bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err
python tools/prof_summary.py $(find $O/p7 -name "*.db" | head -1) > $O/r03_kernel_stats_bench_headline.txt
rm -rf $O/p7
head -6 $O/r03_kernel_stats_bench_headline.txt | cut -c1-200
