#!/usr/bin/env python3
"""zl_msm with HOST scalars at 2^24 (what VariableBaseMSM::multi_scalar_mul is handed): wall time per call, exact check.  ZL_TUNE_HOST_SHARDS sweeps the shard plan.
    python tools/host_msm.py [log_n = 24] [reps = 6]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.init()
from bench import R_BLS, random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381
from openzl_amd.selfcheck import dot_mod_r, expected_point

ln = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << ln
be = Backend(0)
rng = np.random.Generator(np.random.PCG64(5))
k = np.zeros((n, 4), dtype=np.uint64)
k[:, 0] = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
h = be.bases_generate(ZL_BLS12_381, k)
s = random_scalars_lt_r(n, 6)
exp = expected_point(be, ZL_BLS12_381, dot_mod_r(s, k[:, 0].copy(), R_BLS))
d = torch.from_numpy(s.view(np.int64)).cuda()
torch.cuda.synchronize()
for _ in range(2):
    be.msm_dev(h, d.data_ptr(), n)
td = []
for _ in range(3):
    t0 = time.perf_counter(); be.msm_dev(h, d.data_ptr(), n); td.append(time.perf_counter() - t0)
ts = []
for i in range(reps + 1):
    t0 = time.perf_counter()
    xy, inf = be.msm(h, s)
    ts.append(time.perf_counter() - t0)
    assert not inf and (np.asarray(xy) == exp).all()
ts = ts[1:]
print(f"2^{ln} host scalars [{os.environ.get('ZL_TUNE_HOST_SHARDS', 'default')}]: min {min(ts) * 1e3:.2f}  median {np.median(ts) * 1e3:.2f} ms   (device-resident single call: {min(td) * 1e3:.2f} ms)", flush=True)
