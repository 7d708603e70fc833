#!/usr/bin/env python3
"""MSM size / window sweep on the GPU (device-resident inputs): prints ms and points/s per (log_n, c)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254, ZL_G2

if os.environ.get("CURVE", "bls12_381") == "bn254":  # CURVE=bn254: the 8-limb field (32-bit carry-chain multiplier)
    ZL_BLS12_381 = ZL_BN254
    R_BN = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    _rs = random_scalars_lt_r
    random_scalars_lt_r = lambda n, seed: _rs(n, seed, R_BN, 254)  # noqa: E731

be = Backend(0); be.enable_timing(True)
dev = torch.device("cuda", 0)
group = ZL_G2 if "--g2" in sys.argv else 1
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 18, 20, 22, 24]
nmax = 1 << max(sizes)
k = random_scalars_lt_r(nmax, 1); h = be.bases_generate(ZL_BLS12_381, k, group=group)
s = torch.from_numpy(random_scalars_lt_r(nmax, 2).view(np.int64)).to(dev)
pre = [int(x) for x in os.environ.get("PRE", "").split(",") if x]
if pre:
    # precomputed tables: one handle per size (the table covers exactly the points used), c from PRE
    be.bases_free(h)
    for ln in sizes:
        n = 1 << ln
        for c in pre:
            t0 = time.perf_counter()
            hh = be.bases_generate(ZL_BLS12_381, k[:n], group=group)
            t1 = time.perf_counter()
            be.bases_precompute(hh, c)
            t2 = time.perf_counter()
            be.msm_dev(hh, s.data_ptr(), n)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); be.msm_dev(hh, s.data_ptr(), n); ts.append(time.perf_counter() - t0)
            nb = int(os.environ.get("BATCH", "0"))
            if nb:
                be.msm_batch_partial_dev(hh, [s.data_ptr()] * nb, n)
                tb = []
                for _ in range(3):
                    torch.cuda.synchronize(); t0b = time.perf_counter(); be.msm_batch_partial_dev(hh, [s.data_ptr()] * nb, n); tb.append((time.perf_counter() - t0b) / nb)
                tmb = be.last_timing()
                print(f"2^{ln} PRE c={tmb.window_bits:2d} BATCH {nb}: wall/MSM {min(tb)*1e3:8.3f} ms  dev/MSM {tmb.total_ms:8.3f} ms  acc {tmb.dominant_ms:8.3f} ms  {n/min(tb)/1e6:8.1f} Mpts/s", flush=True)
            tm = be.last_timing()
            print(f"2^{ln} PRE c={tm.window_bits:2d}: wall {min(ts)*1e3:8.3f} ms  dev {tm.total_ms:8.3f} ms  acc {tm.dominant_ms:8.3f} ms  {n/min(ts)/1e6:8.1f} Mpts/s   (generate {t1-t0:.2f}s precompute {t2-t1:.2f}s)", flush=True)
            be.bases_free(hh)
    sys.exit(0)
for ln in sizes:
    n = 1 << ln
    for c in [0] + ([int(x) for x in os.environ.get("CS", "").split(",") if x]):
        be.set_msm_window(c)
        be.msm_dev(h, s.data_ptr(), n)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); be.msm_dev(h, s.data_ptr(), n); ts.append(time.perf_counter() - t0)
        tm = be.last_timing()
        print(f"2^{ln} c={tm.window_bits:2d}: wall {min(ts)*1e3:8.3f} ms  dev {tm.total_ms:8.3f} ms  acc {tm.dominant_ms:8.3f} ms  {n/min(ts)/1e6:8.1f} Mpts/s", flush=True)
        nb = int(os.environ.get("BATCH", "0"))
        if nb:
            be.msm_batch_partial_dev(h, [s.data_ptr()] * nb, n)
            tb = []
            for _ in range(3):
                torch.cuda.synchronize(); t0b = time.perf_counter(); be.msm_batch_partial_dev(h, [s.data_ptr()] * nb, n); tb.append((time.perf_counter() - t0b) / nb)
            tmb = be.last_timing()
            print(f"2^{ln} c={tmb.window_bits:2d} BATCH {nb}: wall/MSM {min(tb)*1e3:8.3f} ms  dev/MSM {tmb.total_ms:8.3f} ms  acc {tmb.dominant_ms:8.3f} ms  {n/min(tb)/1e6:8.1f} Mpts/s", flush=True)
