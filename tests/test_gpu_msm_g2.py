"""GPU parity: G2 MSM (Fq2 coordinates; the 5th MSM of every Groth16 prove) vs the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import ZL_G2, ZL_CHECK

pytestmark = pytest.mark.gpu


def _oracle_msm_g2(curve, bases, scalars, threads=8):
    out = np.zeros(4 * ol.nlq(curve), dtype=np.uint64)
    inf = C.c_uint8(0)
    assert ol.lib().zlo_msm_g2(curve.cid, ol.p64(bases), 0, ol.p64(scalars), scalars.shape[0], 0, threads, ol.p64(out), C.byref(inf)) == 0
    return out, inf.value


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 33, 500, 3000])
def test_msm_g2_matches_oracle(backend, curve, n):
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, 300 + n))
    B = gu.g2_mul_gen(curve, ks)
    S = ol.random_scalars(curve, n, 400 + n)
    if n > 10:
        S[0] = 0
        S[1] = ol.ints_to_limbs([1], 4)[0]
        S[2] = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]
        B[3] = 0        # infinity base
        B[5] = B[4]     # doubling inside a bucket
        S[5] = S[4]
    h = backend.bases_upload(curve.cid, B, group=ZL_G2, flags=ZL_CHECK)
    got, inf = backend.msm(h, S)
    assert (backend.bases_download(h, 0, min(n, 8)) == B[: min(n, 8)]).all()
    backend.bases_free(h)
    exp, einf = _oracle_msm_g2(curve, B, S)
    assert inf == einf and (got == exp).all()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_g2_generate_and_known_discrete_log(backend, curve):
    n = 1 << 12
    k = ol.random_scalars(curve, n, 61)
    S = ol.random_scalars(curve, n, 62)
    h = backend.bases_generate(curve.cid, k, group=ZL_G2)
    first = backend.bases_download(h, 0, 4)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    assert (first == gu.g2_mul_gen(curve, ol.limbs_to_ints(k[:4]))).all()
    dot = sum(a * b for a, b in zip(ol.limbs_to_ints(k), ol.limbs_to_ints(S))) % curve.fr.p
    assert inf == 0 and (got == gu.g2_mul_gen(curve, [dot])[0]).all()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("table", [False, True], ids=["plain", "table"])
def test_g2_known_discrete_log_2_18(backend, curve, table):
    """G2 at 2^18 points (long accumulation chains, the c = 16 table mode of a proving key's b_g2_query): exact against (sum s_i k_i mod r) G2.  BN254 G2 runs on the
    lazily reduced 10-limb Fq2 since round 4."""
    n, r = 1 << 18, curve.fr.p
    rng = np.random.Generator(np.random.PCG64(1822))
    k64 = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = k64
    S = ol.random_scalars(curve, n, 1823)
    S[3] = 0
    S[4] = ol.ints_to_limbs([1], 4)[0]
    S[5] = ol.ints_to_limbs([r - 1], 4)[0]
    h = backend.bases_generate(curve.cid, k, group=ZL_G2)
    if table:
        backend.bases_precompute(h, 16)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    s32 = np.ascontiguousarray(S).view(np.uint32).reshape(-1, 8).astype(object)
    dot = 0
    for a in range(8):  # exact dot product in Python integers, one 32-bit column at a time
        dot += int((s32[:, a] * k64.astype(object)).sum()) << (32 * a)
    assert inf == 0 and (got == gu.g2_mul_gen(curve, [dot % r])[0]).all()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [13, 17])
def test_g2_heavy_buckets_known_discrete_log(backend, curve, log_n):
    """Skewed scalars, as a Groth16 witness has them: half of the scalars are ONE value (a bucket per window cut into thousands of chunks: the giant path at 2^17,
    the big path at 2^13), a quarter comes from eight values (big buckets), the rest is uniform.  Since round 6 those buckets are folded by the lane-pair block-tree
    kernels (k_msm_merge_big_pair / _giant_pair / _giant2_pair); exact against (sum s_i k_i mod r) G2, and equal to the one-lane kernels' result."""
    import os

    n, r = 1 << log_n, curve.fr.p
    rng = np.random.Generator(np.random.PCG64(61 + log_n))
    k64 = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = k64
    S = ol.random_scalars(curve, n, 62 + log_n)
    vals = ol.random_scalars(curve, 9, 63)
    who = rng.permutation(n)
    S[who[: n // 2]] = vals[8]
    S[who[n // 2: 3 * n // 4]] = vals[rng.integers(0, 8, size=n // 4)]
    h = backend.bases_generate(curve.cid, k, group=ZL_G2)
    try:
        got, inf = backend.msm(h, S)
        os.environ["ZL_TUNE_G2_PAIR_BLOCKS"] = "0"
        old, old_inf = backend.msm(h, S)
    finally:
        os.environ.pop("ZL_TUNE_G2_PAIR_BLOCKS", None)
        backend.bases_free(h)
    s32 = np.ascontiguousarray(S).view(np.uint32).reshape(-1, 8).astype(object)
    dot = 0
    for a in range(8):
        dot += int((s32[:, a] * k64.astype(object)).sum()) << (32 * a)
    assert inf == 0 and (got == gu.g2_mul_gen(curve, [dot % r])[0]).all()
    assert old_inf == 0 and (old == got).all()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_g2_precomputed_table_and_skew(backend, curve):
    """zl_bases_precompute on a G2 handle (merged bucket set) + the scalar-1 bypass and a giant bucket, against the oracle."""
    n = 2500
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, 71))
    B = gu.g2_mul_gen(curve, ks)
    S = ol.random_scalars(curve, n, 72)
    S[100:700] = ol.ints_to_limbs([1], 4)[0]
    S[700:1500] = ol.ints_to_limbs([5], 4)[0]
    S[1500:1600] = 0
    exp, einf = _oracle_msm_g2(curve, B, S)
    h = backend.bases_upload(curve.cid, B, group=ZL_G2)
    plain, pinf = backend.msm(h, S)
    backend.bases_precompute(h, 16)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    assert pinf == einf and (plain == exp).all()
    assert inf == einf and (got == exp).all()


def test_msm_g2_gls_edge_scalars(backend):
    """BLS12-381 G2 plain MSMs split every scalar into four base-|z| digits over P, psi(P), psi^2(P), psi^3(P) (csrc/zl_msm.hip k_gls_split /
    k_gls_psi; psi acts as [-|z|] on G2).  Scalars on the digit boundaries -- multiples and powers of |z|, all-ones digits, r - 1 -- beside random
    ones, through the LDS sort, the wide sort at c = 16 and a narrow window, single call and pipelined batch, and every edge scalar alone."""
    import torch

    curve = po.BLS12_381
    r = curve.fr.p
    z = 0xD201000000010000
    assert r == z ** 4 - z ** 2 + 1
    edge = [0, 1, 2, z - 1, z, z + 1, z * z - 1, z * z, z * z + 1, z ** 3 - 1, z ** 3, z ** 3 + 1, (z - 1) * (1 + z + z * z + z ** 3) % r, r - 1, r - 2, r - z,
            (z - 1) * z, (z - 1) * z * z, (z - 1) * z ** 3 % r, 3 * z ** 3 + 5, 0xFFFFFFFFFFFFFFFF % z]
    n = 600
    S = ol.random_scalars(curve, n, 881)
    S[: len(edge)] = ol.ints_to_limbs([e % r for e in edge], 4)
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, 882))
    B = gu.g2_mul_gen(curve, ks)
    B[40] = 0  # a base at infinity
    h = backend.bases_upload(curve.cid, B, group=ZL_G2)
    exp, einf = _oracle_msm_g2(curve, B, S)
    d_s = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    try:
        for c in (0, 4, 9, 16):
            backend.set_msm_window(c)
            got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
            assert inf == einf and (got == exp).all(), c
        backend.set_msm_window(0)
        parts = backend.msm_batch_partial_dev(h, [d_s.data_ptr()] * 5, n)
        for j in range(5):
            xy, pinf = backend.partials_sum(curve.cid, parts[j:j + 1], group=ZL_G2)
            assert pinf == einf and (xy == exp).all(), j
        backend.set_msm_window(7)
        for e in edge:
            s1 = ol.ints_to_limbs([e % r], 4)
            got, inf = backend.msm(h, s1)
            e1, i1 = _oracle_msm_g2(curve, B[:1], s1, threads=1)
            assert inf == i1 and (got == e1).all(), hex(e)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_msm_g2_ranges_of_one_handle(backend, curve):
    """Interleaved ranges of one G2 handle: psi / psi^2 / psi^3 images are kept with the handle for the first range used (BLS12-381: GLS) and
    other ranges take the per-call copy; repeated and negated points inside buckets go through the four-lane additions (csrc/zl_quad.h) on Fq2."""
    n = 600
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, 911))
    ks[20:30] = [ks[19]] * 10
    ks[40:45] = [(curve.fr.p - x) % curve.fr.p for x in ks[30:35]]
    B = gu.g2_mul_gen(curve, ks)
    S = ol.random_scalars(curve, n, 912)
    S[20:30] = S[19]
    S[40:45] = S[30:35]
    h = backend.bases_upload(curve.cid, B, group=ZL_G2)
    try:
        for first, cnt in ((1, n - 1), (0, n), (1, n - 1), (5, 300), (0, n)):
            got, inf = backend.msm(h, S[first:first + cnt], first=first)
            exp, einf = _oracle_msm_g2(curve, B[first:first + cnt], S[first:first + cnt])
            assert inf == einf and (got == exp).all(), (first, cnt)
    finally:
        backend.bases_free(h)
