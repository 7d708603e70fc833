#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nttprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
rocprofv3 --kernel-trace --stats -d $O/a -o t -- python tools/ntt_one.py 24 12 > $O/a.log 2>&1
python tools/prof_summary.py $(dbof $O/a) k_ntt_pass > $O/../r04_kernel_stats_ntt_lazy.txt
ZL_NTT_NO_LAZY=1 rocprofv3 --kernel-trace --stats -d $O/b -o t -- python tools/ntt_one.py 24 12 > $O/b.log 2>&1
python tools/prof_summary.py $(dbof $O/b) k_ntt_pass > $O/../r04_kernel_stats_ntt_32bit.txt
rm -rf $O
head -8 $R/gpurun_out/r04_kernel_stats_ntt_lazy.txt | cut -c1-200; tail -12 $R/gpurun_out/r04_kernel_stats_ntt_lazy.txt
head -8 $R/gpurun_out/r04_kernel_stats_ntt_32bit.txt | cut -c1-200; tail -12 $R/gpurun_out/r04_kernel_stats_ntt_32bit.txt
