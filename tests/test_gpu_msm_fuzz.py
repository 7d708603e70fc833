"""Randomised parity sweep of the MSM paths (plain / forced window / precomputed table, uniform and skewed scalars, infinity
bases, duplicated points) against the CPU oracle.  Sizes are small so that the oracle stays fast; seeds are fixed."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po

pytestmark = pytest.mark.gpu


def _case(curve, rng, n):
    k = ol.random_scalars(curve, n, int(rng.integers(1, 1 << 30)))
    B = ol.oracle_g1_mul_gen(curve, k)
    S = ol.random_scalars(curve, n, int(rng.integers(1, 1 << 30)))
    mode = int(rng.integers(0, 4))
    if mode == 1:  # Groth16-witness-like: zeros, ones, small values
        sel = rng.integers(0, 4, size=n)
        S[sel == 0] = 0
        S[sel == 1] = ol.ints_to_limbs([1], 4)[0]
        small = sel == 2
        S[small, 1:] = 0
        S[small, 0] &= np.uint64(0xFFFF)
    elif mode == 2:  # few distinct scalars -> giant buckets
        S[:] = S[rng.integers(0, 3, size=n)]
    elif mode == 3:  # few distinct points, infinity bases
        B[:] = B[rng.integers(0, max(1, n // 8), size=n)]
        B[rng.integers(0, n, size=max(1, n // 10))] = 0
    return B, S


@pytest.mark.parametrize("seed", range(int(os.environ.get("ZL_FUZZ_SEEDS", "12"))))  # ZL_FUZZ_SEEDS=200 for a longer soak
def test_msm_fuzz(backend, seed):
    rng = np.random.default_rng(1000 + seed)
    curve = po.BLS12_381 if seed % 3 else po.BN254
    n = int(rng.integers(1, 6000))
    B, S = _case(curve, rng, n)
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    h = backend.bases_upload(curve.cid, B)
    try:
        got, inf = backend.msm(h, S)
        assert inf == einf and (got == exp).all(), "plain"
        c = int(rng.integers(2, 17))
        backend.set_msm_window(c)
        got, inf = backend.msm(h, S)
        backend.set_msm_window(0)
        assert inf == einf and (got == exp).all(), f"window {c}"
        pc = int(rng.integers(16, 22))
        backend.bases_precompute(h, pc)
        got, inf = backend.msm(h, S)
        assert inf == einf and (got == exp).all(), f"table {pc}"
        lo = int(rng.integers(0, n))
        hi = int(rng.integers(lo, n + 1))
        got, inf = backend.msm(h, S[lo:hi], first=lo)
        e2, e2inf = ol.oracle_msm_g1(curve, B[lo:hi], S[lo:hi], algo=0, threads=8)
        assert inf == e2inf and (got == e2).all(), "table sub-range"
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)
