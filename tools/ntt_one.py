#!/usr/bin/env python3
"""A few 2^log_n forward / inverse NTTs (Montgomery in / out, device-resident), for clean rocprofv3 kernel stats / PMC passes.
    python tools/ntt_one.py [log_n = 24] [reps = 3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381

ln = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
be = Backend(0)
be.enable_timing(True)
x = random_scalars_lt_r(1 << ln, 3000)
dx = torch.from_numpy(x.view(np.int64)).cuda()
torch.cuda.synchronize()
for _ in range(reps):
    be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=False, mont=True)
    f = be.last_timing().total_ms
    be.ntt_dev(ZL_BLS12_381, dx.data_ptr(), ln, inverse=True, mont=True)
    print(f"2^{ln}: forward {f:.3f} ms  inverse {be.last_timing().total_ms:.3f} ms", flush=True)
assert (dx.cpu().numpy().view(np.uint64) == x).all()
