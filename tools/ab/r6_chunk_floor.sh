#!/bin/bash
# round 6: chunk length floored by the bucket density (ZL_CHUNK >= entries per bucket / 4) -- the 2^14 G2 anomaly
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_chunk_floor.log; : > $L
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_gpu_msm_g2.py tests/test_groth16.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 >> $L
for rep in 1 2; do
  echo "== G2 bls" >> $L; python tools/msm_sweep.py --g2 8 10 11 12 13 14 15 16 17 18 2>&1 | grep "2^" >> $L
  echo "== G2 bn254" >> $L; CURVE=bn254 python tools/msm_sweep.py --g2 10 12 13 14 15 16 2>&1 | grep "2^" >> $L
  echo "== G1 bls" >> $L; python tools/msm_sweep.py 10 12 13 14 15 16 17 18 2>&1 | grep "2^" >> $L
  echo "== G1 bn254" >> $L; CURVE=bn254 python tools/msm_sweep.py 10 12 14 16 18 2>&1 | grep "2^" >> $L
  for k in 1 8 64 256; do ITERS=20 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
done
cat $L
