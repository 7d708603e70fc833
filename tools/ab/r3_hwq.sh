#!/bin/bash
# does the number of HSA queues change the side-by-side small jobs?  + kernel trace of the k = 1 proof (queue ids per launch)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3hwq}
mkdir -p $O
cd $R
for q in 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/small_lat.py g16 2>&1 | grep "^Groth"
done
echo "ZL_TUNE_LANE_THREADS=0"; ZL_TUNE_LANE_THREADS=0 python tools/small_lat.py g16 2>&1 | grep "^Groth"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o k1 -- python $R/tools/small_lat.py g16 > $O/prof.log 2>&1
cd $R
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python tools/timeline.py $DB 300 2 > $O/timeline_g16_k1.txt 2>&1
head -150 $O/timeline_g16_k1.txt | cut -c1-120
