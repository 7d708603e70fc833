#!/usr/bin/env python3
"""round 6: G2 MSM with a SKEWED scalar vector (a quarter of the scalars take one of 8 values: heavy buckets -> merge_big / merge_giant), ZL_TUNE_G2_PAIR_BLOCKS=0/1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381, ZL_G2
be = Backend(0); be.enable_timing(True)
for ln in (16, 18, 20):
    n = 1 << ln
    k = random_scalars_lt_r(n, 1); h = be.bases_generate(ZL_BLS12_381, k, group=ZL_G2)
    s = random_scalars_lt_r(n, 2)
    vals = random_scalars_lt_r(8, 3)
    rng = np.random.Generator(np.random.PCG64(4))
    idx = rng.integers(0, n, size=n // 4)
    s[idx] = vals[rng.integers(0, 8, size=n // 4)]
    d = torch.from_numpy(s.view(np.int64)).cuda()
    ref = None
    for v in ("0", "1", "0", "1"):
        os.environ["ZL_TUNE_G2_PAIR_BLOCKS"] = v
        be.msm_dev(h, d.data_ptr(), n)
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); out = be.msm_dev(h, d.data_ptr(), n); ts.append(time.perf_counter() - t0)
        if ref is None: ref = out
        assert (np.asarray(out[0]) == np.asarray(ref[0])).all()
        print(f"2^{ln} skewed G2, pair blocks {v}: wall {min(ts)*1e3:.3f} ms  dev {be.last_timing().total_ms:.3f}", flush=True)
    be.bases_free(h)
