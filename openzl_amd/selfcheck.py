"""Known-discrete-log self check for full-size MSMs: with bases P_i = k_i * G the exact answer of
MSM(s, P) is (sum_i s_i k_i mod r) * G -- one O(n) dot product on the host and ONE scalar multiplication, no CPU MSM.
numpy only; used by bench.py's correctness gate and by tests/ at 2^24 / 2^26."""
from __future__ import annotations

import numpy as np


def dot_mod_r(S: np.ndarray, K: np.ndarray, r: int) -> int:
    """sum_i S_i * K_i mod r, exact, for (n, 4) uint64 little-endian limb arrays S and K (K may also be (n,) uint64).
    32-bit limb products are split into halves so that up to 2^31 of them sum inside a uint64; row ranges on host threads
    (numpy releases the GIL), in blocks that stay in cache."""
    import os
    from concurrent.futures import ThreadPoolExecutor

    if S.shape[0] == 0:
        return 0
    s32 = np.ascontiguousarray(S).view(np.uint32).reshape(S.shape[0], -1)
    K = np.ascontiguousarray(K)
    k32 = K.view(np.uint32).reshape(K.shape[0], -1)
    n, ns, nk = s32.shape[0], s32.shape[1], k32.shape[1]
    assert n == k32.shape[0] and n < (1 << 31)
    m32 = np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    BLK = 1 << 15
    live = [b for b in range(nk) if k32[:, b].any()]

    def rows(lo_row: int, hi_row: int) -> int:
        acc = [0] * (ns + nk)
        for i in range(lo_row, hi_row, BLK):
            j = min(i + BLK, hi_row)
            sb = s32[i:j].astype(np.uint64)
            for b in live:
                kb = k32[i:j, b].astype(np.uint64)
                for a in range(ns):
                    prod = sb[:, a] * kb
                    acc[a + b] += int((prod & m32).sum(dtype=np.uint64)) + (int((prod >> sh).sum(dtype=np.uint64)) << 32)
        return sum(v << (32 * w) for w, v in enumerate(acc))

    try:
        cpus = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        cpus = os.cpu_count() or 1
    workers = max(1, min(32, cpus, (n + BLK - 1) // BLK))
    step = (n + workers - 1) // workers
    with ThreadPoolExecutor(max_workers=workers) as pool:
        total = sum(pool.map(lambda lo: rows(lo, min(lo + step, n)), range(0, n, step)))
    return total % r


def expected_point(be, curve_id: int, dot: int) -> np.ndarray:
    """canonical affine x||y of dot * G, computed by the device generator (one point)"""
    kd = np.array([[(dot >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
    hd = be.bases_generate(curve_id, kd)
    try:
        return be.bases_download(hd)[0]
    finally:
        be.bases_free(hd)
