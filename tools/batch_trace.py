#!/usr/bin/env python3
"""One warm pipelined batch of MSMs (zl_msm_batch_partial_dev) for a rocprofv3 kernel trace: python tools/batch_trace.py <log_n> <count>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381
ln, cnt = int(sys.argv[1]), int(sys.argv[2])
n = 1 << ln
be = Backend(0); be.enable_timing(True)
k = np.zeros((n, 4), dtype=np.uint64); k[:, 0] = np.random.Generator(np.random.PCG64(1)).integers(1, 1 << 63, size=n, dtype=np.uint64)
h = be.bases_generate(ZL_BLS12_381, k)
d = [torch.from_numpy(random_scalars_lt_r(n, 2 + j).view(np.int64)).cuda() for j in range(2)]
torch.cuda.synchronize()
be.msm_batch_partial_dev(h, [d[i % 2].data_ptr() for i in range(3)], n)
torch.cuda.synchronize(); time.sleep(0.05)
t0 = time.perf_counter(); be.msm_batch_partial_dev(h, [d[i % 2].data_ptr() for i in range(cnt)], n); dt = time.perf_counter() - t0
tm = be.last_timing()
print(f"2^{ln} batch {cnt}: {dt*1e3/cnt:.3f} ms/MSM (device {tm.total_ms:.3f}, acc {tm.dominant_ms:.3f}) c={tm.window_bits}")
