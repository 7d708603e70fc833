#!/bin/bash
# round 6: measured window tables on (1) against the cost model (0): G2 sizes and the proofs that contain them
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_quarter_table_ab.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm_g2.py tests/test_gpu_msm_fuzz.py tests/test_groth16.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -3 >> $L
for rep in 1 2 3; do
for v in 0 1; do
  echo "== ZL_TUNE_HALF_TABLE=$v" >> $L
  ZL_TUNE_HALF_TABLE=$v python tools/msm_sweep.py --g2 10 12 13 14 15 16 18 2>&1 | grep -v amdgpu.ids | sed 's/^/g2 /' >> $L
  for k in 1 8 64 256; do ZL_TUNE_HALF_TABLE=$v ITERS=30 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
done
done
cat $L
