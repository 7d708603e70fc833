"""Known-discrete-log self check for full-size MSMs: with bases P_i = k_i * G the exact answer of
MSM(s, P) is (sum_i s_i k_i mod r) * G -- one O(n) dot product on the host and ONE scalar multiplication, no CPU MSM.
numpy only; used by bench.py's correctness gate and by tests/ at 2^24 / 2^26."""
from __future__ import annotations

import numpy as np


def dot_mod_r(S: np.ndarray, K: np.ndarray, r: int) -> int:
    """sum_i S_i * K_i mod r, exact, for (n, 4) uint64 little-endian limb arrays S and K (K may also be (n,) uint64).
    32-bit limb products are split into halves so that up to 2^31 of them sum inside a uint64."""
    s32 = np.ascontiguousarray(S).view(np.uint32).reshape(S.shape[0], -1)
    K = np.ascontiguousarray(K)
    k32 = K.view(np.uint32).reshape(K.shape[0], -1)
    assert s32.shape[0] == k32.shape[0] and s32.shape[0] < (1 << 31)
    total = 0
    m32 = np.uint64(0xFFFFFFFF)
    sh = np.uint64(32)
    for b in range(k32.shape[1]):
        kb = k32[:, b].astype(np.uint64)
        if not kb.any():
            continue
        for a in range(s32.shape[1]):
            prod = s32[:, a].astype(np.uint64) * kb
            lo = int((prod & m32).sum(dtype=np.uint64))
            hi = int((prod >> sh).sum(dtype=np.uint64))
            total += (lo + (hi << 32)) << (32 * (a + b))
    return total % r


def expected_point(be, curve_id: int, dot: int) -> np.ndarray:
    """canonical affine x||y of dot * G, computed by the device generator (one point)"""
    kd = np.array([[(dot >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
    hd = be.bases_generate(curve_id, kd)
    try:
        return be.bases_download(hd)[0]
    finally:
        be.bases_free(hd)
