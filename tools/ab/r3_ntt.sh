#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3ntt}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_sharded_ntt.py tests/test_groth16.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log
echo "== table"; python tools/ntt_one.py 24 12 2>&1 | tail -4; python tools/ntt_one.py 20 12 2>&1 | tail -2
echo "== ZL_NTT_NO_LAST_TABLE=1"; ZL_NTT_NO_LAST_TABLE=1 python tools/ntt_one.py 24 12 2>&1 | tail -4; ZL_NTT_NO_LAST_TABLE=1 python tools/ntt_one.py 20 12 2>&1 | tail -2
