#!/bin/bash
# round 6: the block-tree kernels of the heavy buckets (merge_big / giant / giant2) of the Fq2 groups on lane pairs (ZL_TUNE_G2_PAIR_BLOCKS=1) against the one-lane kernels
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_pair_blocks_ab.log; : > $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $L
timeout 1500 python -m pytest tests/test_gpu_msm_g2.py tests/test_gpu_msm_fuzz.py tests/test_groth16.py tests/test_gpu_lanes.py tests/test_gpu_multi.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_G2_PAIR_BLOCKS=$v" >> $L
  ZL_TUNE_G2_PAIR_BLOCKS=$v ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  ZL_TUNE_G2_PAIR_BLOCKS=$v CURVE=bn254 ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  ZL_TUNE_G2_PAIR_BLOCKS=$v ITERS=30 python tools/g16_one.py 64 2>&1 | grep "prove k=" >> $L
done
done
cat $L
