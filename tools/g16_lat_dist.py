#!/usr/bin/env python3
"""Distribution of small-proof latencies (every sample printed): looks for outliers behind a good median.
    python tools/g16_lat_dist.py [k=1] [iters=60]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.init()
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254, Circuit, Groth16Keys

k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
be = Backend(0)
be.enable_timing(True)
if os.environ.get("PRE_LEGS"):
    from pre_legs import run_pre_legs
    run_pre_legs(be)
circ = Circuit(ZL_BN254 if os.environ.get("CURVE") == "bn254" else ZL_BLS12_381, k)
keys = Groth16Keys(be, circ, seed=1)
ts = []
for _ in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keys.prove(seed=3)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"k={k}: " + " ".join(f"{t:.2f}" for t in ts))
a = np.array(ts[3:])
print(f"k={k}: after 3 warm-ups: min {a.min():.3f} median {np.median(a):.3f} mean {a.mean():.3f} p90 {np.percentile(a, 90):.3f} max {a.max():.3f} ms")
