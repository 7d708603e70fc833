#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nttprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
rocprofv3 --kernel-trace --stats -d $O/a -o t -- python tools/ntt_one.py 24 12 > $O/a.log 2>&1
python tools/prof_summary.py $(dbof $O/a) k_ntt_pass > $O/../r04_kernel_stats_ntt_lazy.txt
rm -rf $O
head -6 $R/gpurun_out/r04_kernel_stats_ntt_lazy.txt | cut -c1-200; tail -9 $R/gpurun_out/r04_kernel_stats_ntt_lazy.txt
