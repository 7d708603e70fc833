#!/usr/bin/env python3
"""A few single-call MSMs of one configuration, for `rocprofv3 --kernel-trace --stats` (clean per-kernel durations).
    python tools/msm_one.py <log_n> <c: 0 = auto> [pre_c: -1 = no table] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381

log_n, c = int(sys.argv[1]), int(sys.argv[2])
pre = int(sys.argv[3]) if len(sys.argv) > 3 else -1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
be = Backend(0)
be.enable_timing(True)
n = 1 << log_n
rng = np.random.Generator(np.random.PCG64(1))
k = np.zeros((n, 4), dtype=np.uint64)
k[:, 0] = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
h = be.bases_generate(ZL_BLS12_381, k)
if pre >= 0:
    be.bases_precompute(h, pre)
s = torch.from_numpy(random_scalars_lt_r(n, 2).view(np.int64)).cuda()
torch.cuda.synchronize()
if c:
    be.set_msm_window(c)
for _ in range(reps):
    t0 = time.perf_counter()
    be.msm_dev(h, s.data_ptr(), n)
    dt = time.perf_counter() - t0
    tm = be.last_timing()
    print(f"2^{log_n} c={tm.window_bits} pre={pre}: wall {dt*1e3:.3f} ms dev {tm.total_ms:.3f} acc {tm.dominant_ms:.3f}", flush=True)
