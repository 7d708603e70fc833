#!/usr/bin/env python3
"""Does a small proof slow down after other GPU work of the same process?  (a) fresh; (b) after 1 s of torch matmuls (no library state);
(c) after single-call MSMs of this library up to 2^22; (d) after (c) + closing and re-creating the keys."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381, Circuit, Groth16Keys
be = Backend(0)
circ = Circuit(ZL_BLS12_381, 1)
keys = Groth16Keys(be, circ, seed=1)
def t(tag, keys):
    for _ in range(3): keys.prove(seed=3)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); keys.prove(seed=3); ts.append(time.perf_counter() - t0)
    print(f"{tag}: min {min(ts)*1e3:.3f} med {np.median(ts)*1e3:.3f} ms", flush=True)
t("fresh", keys)
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 1.0:
    b = a @ a
torch.cuda.synchronize()
t("after 1 s of bf16 matmuls", keys)
n = 1 << 22
k = np.zeros((n, 4), dtype=np.uint64); k[:, 0] = np.random.Generator(np.random.PCG64(1)).integers(1, 1 << 63, size=n, dtype=np.uint64)
h = be.bases_generate(ZL_BLS12_381, k)
s = torch.from_numpy(random_scalars_lt_r(n, 2).view(np.int64)).cuda(); torch.cuda.synchronize()
for ln in (16, 20, 22):
    for _ in range(5): be.msm_dev(h, s.data_ptr(), 1 << ln)
t("after single-call MSMs up to 2^22", keys)
be.bases_free(h)
keys.close()
keys = Groth16Keys(be, circ, seed=1)
t("after re-creating the keys", keys)
