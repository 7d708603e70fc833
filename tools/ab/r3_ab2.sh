#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 6 --warmup 2 --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('msm', round(d['ms_per_step'],2), 'g16', round(d['groth16']['prove_ms'],2), 'ntt', round(d['ntt']['forward_ms'],3))"; }
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8 ZL_TUNE_QUAD_LANES=0 ZL_TUNE_QUAD_ACC_CHUNKS=0
run GPU_MAX_HW_QUEUES=4 ZL_TUNE_QUAD_LANES=0 ZL_TUNE_QUAD_ACC_CHUNKS=0
run GPU_MAX_HW_QUEUES=8 ZL_TUNE_ENDO_CACHE_MB=0
run GPU_MAX_HW_QUEUES=8
