#!/usr/bin/env python3
"""Latency of the small / mid-size calls: 2^16 BN254 + BLS12-381, 2^18, 2^20, 2^22 single MSM calls (wall, device, accumulation)
and Groth16 proofs of the k = 1 / 64 circuits.  ZL_HOST_TRACE=1 adds the host-side phase times of every single MSM call on stderr.
    python tools/small_lat.py [msm] [g16]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254, Circuit, Groth16Keys

R_BN = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
what = sys.argv[1:] or ["msm", "g16"]
be = Backend(0)
be.enable_timing(True)
if "msm" in what:
    sizes = tuple(int(x) for x in os.environ.get("SIZES", "16,18,20,22").split(","))
    for curve, name, logs in ((ZL_BN254, "bn254", (16,)), (ZL_BLS12_381, "bls12_381", sizes)):
        nmax = 1 << max(logs)
        rng = np.random.Generator(np.random.PCG64(1))
        k = np.zeros((nmax, 4), dtype=np.uint64)
        k[:, 0] = rng.integers(1, 1 << 63, size=nmax, dtype=np.uint64)
        h = be.bases_generate(curve, k)
        sc = random_scalars_lt_r(nmax, 2) if curve == ZL_BLS12_381 else random_scalars_lt_r(nmax, 2, R_BN, 254)
        s = torch.from_numpy(sc.view(np.int64)).cuda()
        torch.cuda.synchronize()
        for ln in logs:
            n = 1 << ln
            for _ in range(3):
                be.msm_dev(h, s.data_ptr(), n)
            ts, dv, ac = [], [], []
            for _ in range(10):
                t0 = time.perf_counter()
                be.msm_dev(h, s.data_ptr(), n)
                ts.append(time.perf_counter() - t0)
                tm = be.last_timing()
                dv.append(tm.total_ms)
                ac.append(tm.dominant_ms)
            print(f"MSM {name} 2^{ln} c={tm.window_bits}: wall min {min(ts)*1e3:.3f} med {np.median(ts)*1e3:.3f} ms  dev {np.median(dv):.3f}  acc {np.median(ac):.3f}", flush=True)
        be.bases_free(h)
if "g16" in what:
    for kk in (1, 64):
        circ = Circuit(ZL_BLS12_381, kk)
        keys = Groth16Keys(be, circ, seed=1)
        for _ in range(3):
            keys.prove(seed=3)
        ts = []
        for _ in range(int(os.environ.get('ITERS', '10'))):
            t0 = time.perf_counter()
            keys.prove(seed=3)
            ts.append(time.perf_counter() - t0)
        print(f"Groth16 k={kk} ({circ.shape[0]} constraints): prove min {min(ts)*1e3:.3f} med {np.median(ts)*1e3:.3f} ms (device {be.last_timing().total_ms:.3f})", flush=True)
        keys.close()
        circ.close()
