#!/bin/bash
# quick validation: MSM / Groth16 parity tests, small-call latencies, host-scalar path A/B
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3quick}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_groth16.py tests/test_gpu_abi_errors.py tests/test_gpu_msm_g2.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
ZL_HOST_TRACE=1 python tools/small_lat.py > $O/small_lat.log 2>&1
grep -v "zl_msm n=\|amdgpu" $O/small_lat.log
python - > $O/host_scalars.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381
be = Backend(0)
for ln in (22, 24):
    n = 1 << ln
    k = np.zeros((n, 4), dtype=np.uint64); k[:, 0] = np.random.Generator(np.random.PCG64(1)).integers(1, 1 << 63, size=n, dtype=np.uint64)
    h = be.bases_generate(ZL_BLS12_381, k)
    s = random_scalars_lt_r(n, 2)
    d = torch.from_numpy(s.view(np.int64)).cuda(); torch.cuda.synchronize()
    ref = be.msm_dev(h, d.data_ptr(), n)
    for env in ("", "1"):
        if env: os.environ["ZL_NO_HOST_CHUNKS"] = "1"
        else: os.environ.pop("ZL_NO_HOST_CHUNKS", None)
        be.msm(h, s)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); out = be.msm(h, s); ts.append(time.perf_counter() - t0)
        assert (out[0] == ref[0]).all()
        print(f"2^{ln} host scalars {'one copy + one MSM' if env else 'growing shards   '}: min {min(ts)*1e3:.2f} ms  med {np.median(ts)*1e3:.2f} ms", flush=True)
    t0 = time.perf_counter(); be.msm_dev(h, d.data_ptr(), n); print(f"2^{ln} device scalars single call {(time.perf_counter()-t0)*1e3:.2f} ms")
    be.bases_free(h)
PY
cat $O/host_scalars.log | grep -v amdgpu
