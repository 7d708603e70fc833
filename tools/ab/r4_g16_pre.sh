#!/bin/bash
# does what ran before in the process change the large proof?  (bench.py's Groth16 leg is ~1 ms slower than tools/g16_one.py on the same box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/g16pre; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
dbof() { find $1 -name "*.db" | head -1; }
for rep in 1 2; do
for pre in 0 1; do
  if [ $pre = 1 ]; then export PRE_LEGS=1; else unset PRE_LEGS; fi
  echo "== PRE_LEGS=$pre" >> $O/log.txt
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $O/log.txt
done
done
export PRE_LEGS=1
rocprofv3 --kernel-trace --stats -d $O/p5 -o t -- python tools/g16_one.py > $O/g16_one.log 2>&1
python tools/timeline.py $(dbof $O/p5) 1500 1 150 > $O/g16_timeline_pre.txt 2>&1
rm -rf $O/p5
cat $O/log.txt
