#!/usr/bin/env python3
"""A few Groth16 proofs of the config-5 circuit, for a rocprofv3 kernel trace (tools/g16_timeline.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
from openzl_amd import Backend, ZL_BLS12_381, Circuit, Groth16Keys

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
be = Backend(0)
be.enable_timing(True)
t0 = time.perf_counter(); circ = Circuit(ZL_BLS12_381, k); t1 = time.perf_counter()
keys = Groth16Keys(be, circ, seed=1); t2 = time.perf_counter()
print(f"synthesis {t1 - t0:.2f} s  setup {t2 - t1:.2f} s", flush=True)
for _ in range(4):
    t0 = time.perf_counter()
    keys.prove(seed=3)
    print(f"prove {(time.perf_counter() - t0) * 1e3:.2f} ms (device {be.last_timing().total_ms:.2f})", flush=True)
