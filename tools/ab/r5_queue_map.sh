#!/bin/bash
# which hardware queues the chains of a 14 977-constraint proof land on, by small-lane class and by what ran before (rocprofv3 queue ids of the last proof)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pre in 0 1; do for prio in 0 2 3; do
  d=/tmp/qm_${pre}_${prio}; rm -rf $d
  ( [ $pre = 1 ] && export PRE_LEGS=1; ZL_TUNE_SMALL_LANE_PRIO=$prio ITERS=12 rocprofv3 --kernel-trace --stats -d $d -o t -- python tools/g16_one.py 64 > $d.log 2>&1 )
  echo "=== PRE_LEGS=$pre SMALL_LANE_PRIO=$prio: $(grep 'prove k=' $d.log)"
  python tools/queue_map.py $(find $d -name "*.db" | head -1)
done; done
