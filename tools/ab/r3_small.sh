#!/bin/bash
# small-call latencies + host trace of the k = 1 proof + Groth16 parity tests
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3small}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_key_wire.py tests/test_pairing_verify.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
SIZES=16,20 python tools/small_lat.py 2>&1 | grep -v amdgpu | tee $O/small_lat.log
ZL_HOST_TRACE=1 python tools/small_lat.py g16 2>&1 | grep "zl_msm jobs\|zl_groth16\|^Groth\|zl_msm n=" > $O/g16_trace.log
python - $O/g16_trace.log <<'PY'
import sys
lines = open(sys.argv[1]).read().splitlines()
idx = [i for i, l in enumerate(lines) if 'nc=235]' in l and 'z on device' in l][-1]
print("\n".join(lines[idx:idx + 26]))
PY
