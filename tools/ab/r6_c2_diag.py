#!/usr/bin/env python3
"""round 6: why does bench.py's config-2 leg see 3.32 ms per pipelined 2^20 MSM where tools/msm_sweep.py sees 2.96 for the same batch of 6?
Times the batch (a) as the sweep does, (b) one shot, (c) with two alternating scalar vectors, (d) through sharded_msm_batch (host folds inside the timed region),
each in a fresh ctx and again after a 2^24 batch + a host-scalar MSM in the same process (PRE=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381
from openzl_amd.sharded import sharded_msm_batch

be = Backend(0); be.enable_timing(True)
dev = torch.device("cuda", 0)
n = 1 << 20
if os.environ.get("PRE") == "1":
    N = 1 << 24
    kk = random_scalars_lt_r(N, 5); hh = be.bases_generate(ZL_BLS12_381, kk)
    ss = torch.from_numpy(random_scalars_lt_r(N, 6).view(np.int64)).to(dev)
    be.msm_batch_partial_dev(hh, [ss.data_ptr()] * 4, N)
    be.msm(hh, random_scalars_lt_r(N, 7))
    be.bases_free(hh); del ss
k = random_scalars_lt_r(n, 1); h = be.bases_generate(ZL_BLS12_381, k)
s0 = torch.from_numpy(random_scalars_lt_r(n, 2).view(np.int64)).to(dev)
s1 = torch.from_numpy(random_scalars_lt_r(n, 3).view(np.int64)).to(dev)
be.msm_dev(h, s0.data_ptr(), n)
be.msm_batch_partial_dev(h, [s0.data_ptr()] * 3, n)
def t(fn, reps):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); out.append((time.perf_counter() - t0) * 1e3)
    return out
for nb in (6, 20):
    a = t(lambda: be.msm_batch_partial_dev(h, [s0.data_ptr()] * nb, n), 5)
    tm = be.last_timing()
    b = t(lambda: be.msm_batch_partial_dev(h, [(s0, s1)[i % 2].data_ptr() for i in range(nb)], n), 5)
    c = t(lambda: sharded_msm_batch(be.msm_batch_partial_dev(h, [(s0, s1)[i % 2].data_ptr() for i in range(nb)], n), ZL_BLS12_381), 5)
    f = lambda v: " ".join(f"{x / nb:.3f}" for x in v)
    print(f"batch {nb}: same vector [{f(a)}]  alternating [{f(b)}]  + host folds [{f(c)}]  ms/MSM   (dev/MSM {tm.total_ms:.3f})", flush=True)
single = t(lambda: be.msm_partial_dev(h, s0.data_ptr(), n), 10)
print("single call:", " ".join(f"{x:.3f}" for x in single), " dev", be.last_timing().total_ms, "acc", be.last_timing().dominant_ms)
