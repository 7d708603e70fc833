#!/bin/bash
# round 6: timelines of a 2^14 BLS12-381 G1 MSM with and without the endomorphism split (the split is slower there: why?)
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
dbof() { find $1 -name "*.db" | head -1; }
for ln in 14 16; do
rocprofv3 --kernel-trace --stats -d $O/tg$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/tl_glv_$ln.log 2>&1
python tools/timeline.py $(dbof $O/tg$ln) 150 > $O/r06_timeline_msm_2_${ln}_glv.txt 2>&1
ZL_NO_GLV=1 rocprofv3 --kernel-trace --stats -d $O/tp$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/tl_plain_$ln.log 2>&1
python tools/timeline.py $(dbof $O/tp$ln) 150 > $O/r06_timeline_msm_2_${ln}_plain.txt 2>&1
rm -rf $O/tg$ln $O/tp$ln
done
