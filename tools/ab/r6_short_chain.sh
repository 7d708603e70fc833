#!/bin/bash
# round 6: shorter launch chains of small MSMs (one zeroing kernel instead of a three-kernel memset, heavy-bucket kernels skipped when no bucket can be heavy, k_msm_ones finishes its own sum,
# one result copy; then level 0 + the whole tree in one launch for tiny jobs)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_short_chain.log; : > $L
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
for rep in 1 2 3; do
  for k in 1 8 64; do ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $L
done
python tools/msm_sweep.py 8 10 12 14 16 2>&1 | grep "2^" >> $L
python tools/msm_sweep.py --g2 8 10 12 14 16 2>&1 | grep "2^" >> $L
CURVE=bn254 python tools/msm_sweep.py 10 14 16 2>&1 | grep "2^" >> $L
ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
cat $L
