#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3glv}
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_fuzz.py tests/test_groth16.py tests/test_gpu_abi_errors.py tests/test_gpu_multi.py tests/test_host_mirror.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -n "passed\|failed\|Error\|error" $O/pytest.log | head
python tools/small_lat.py > $O/small_lat_glv.log 2>&1; grep "^MSM\|^Groth" $O/small_lat_glv.log
ZL_NO_GLV=1 python tools/small_lat.py msm > $O/small_lat_noglv.log 2>&1; echo "--- ZL_NO_GLV=1"; grep "^MSM" $O/small_lat_noglv.log
BATCH=6 python tools/msm_sweep.py 20 22 24 > $O/sweep_glv.log 2>&1; grep "^2\^" $O/sweep_glv.log
echo "--- ZL_NO_GLV=1"; ZL_NO_GLV=1 BATCH=6 python tools/msm_sweep.py 24 > $O/sweep_noglv.log 2>&1; grep "^2\^" $O/sweep_noglv.log
