#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3bt}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
for mode in glv noglv; do
  if [ $mode = noglv ]; then export ZL_NO_GLV=1; else unset ZL_NO_GLV; fi
  rocprofv3 --kernel-trace --stats -d $O/p_$mode -o t -- python tools/batch_trace.py 24 5 > $O/batch_$mode.log 2>&1
  python tools/timeline.py $(dbof $O/p_$mode) 2000 1 150 > $O/timeline_batch_$mode.txt 2>&1
  grep "batch" $O/batch_$mode.log
done
rm -rf $O/p_glv $O/p_noglv
cat $O/timeline_batch_noglv.txt
