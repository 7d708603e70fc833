#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3persist}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msm.py -m gpu -x -q -k "2_24 or pipelined or adversarial or glv" > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log
for WG in 0 2 3; do
  echo "== ZL_TUNE_ACC_WG_PER_CU=$WG"
  ZL_TUNE_ACC_WG_PER_CU=$WG python tools/batch_trace.py 24 8 2>&1 | grep batch
  ZL_TUNE_ACC_WG_PER_CU=$WG python tools/batch_trace.py 24 8 2>&1 | grep batch
done
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/p -o t -- python tools/batch_trace.py 24 5 > $O/batch.log 2>&1
python tools/timeline.py $(find $O/p -name "*.db" | head -1) 2000 1 150 > $O/timeline_batch_persist.txt 2>&1
rm -rf $O/p
head -40 $O/timeline_batch_persist.txt
