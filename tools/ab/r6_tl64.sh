#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 8 64; do
rocprofv3 --kernel-trace --stats -d $O/pg$k -o t -- python tools/g16_one.py $k > $O/g16_one_$k.log 2>&1
python tools/timeline.py $(find $O/pg$k -name "*.db" | head -1) 150 > $O/r06_timeline_g16_k${k}_batched.txt 2>&1
rm -rf $O/pg$k
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py $k 2>&1 | grep -v "prove \|synth\|amdgpu.ids" | tail -19 > $O/r06_trace_k${k}_batched.txt
done
