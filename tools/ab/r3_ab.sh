#!/bin/bash
# A/B of the host-side knobs for small proofs / small MSMs
set -u
R=$GRAFT_REPO_ROOT
cd $R
for q in 4 8; do for lt in 0 1; do
  echo "== GPU_MAX_HW_QUEUES=$q ZL_TUNE_LANE_THREADS=$lt"
  GPU_MAX_HW_QUEUES=$q ZL_TUNE_LANE_THREADS=$lt ITERS=40 python tools/small_lat.py g16 2>&1 | grep "^Groth"
done; done
echo "== quad off (queues 8, lane threads 0)"
GPU_MAX_HW_QUEUES=8 ZL_TUNE_LANE_THREADS=0 ZL_TUNE_QUAD_LANES=0 ZL_TUNE_QUAD_ACC_CHUNKS=0 ITERS=40 python tools/small_lat.py g16 2>&1 | grep "^Groth"
for q in 4 8; do
  echo "== MSM sizes, GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q SIZES=16,18,20,22 python tools/small_lat.py msm 2>&1 | grep "^MSM"
done
echo "== quad lanes sweep at 2^16 / 2^20 (queues 8)"
for ql in 0 16384 65536 262144; do echo "ZL_TUNE_QUAD_LANES=$ql"; GPU_MAX_HW_QUEUES=8 ZL_TUNE_QUAD_LANES=$ql SIZES=16,20 python tools/small_lat.py msm 2>&1 | grep "^MSM bls"; done
for q in 4 8; do
  echo "== headline, GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 10 --warmup 2 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('groth16',{}).get('prove_ms'), d.get('ntt',{}).get('ms_per_transform'))"
done
