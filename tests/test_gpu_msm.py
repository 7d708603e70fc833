"""GPU parity: zl_msm (HIP Pippenger) vs the CPU oracle, bit-exact on canonical affine coordinates."""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd import ZL_G1, ZL_CHECK, ZL_MONT, BackendError

pytestmark = pytest.mark.gpu
CURVES = [po.BLS12_381, po.BN254]


def _bases(curve, n, seed):
    k = ol.random_scalars(curve, n, seed)
    return k, ol.oracle_g1_mul_gen(curve, k)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 100, 1000, 4097])
def test_msm_matches_oracle(backend, curve, n):
    k, B = _bases(curve, n, 100 + n)
    S = ol.random_scalars(curve, n, 200 + n)
    h = backend.bases_upload(curve.cid, B)
    got, inf = backend.msm(h, S)
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    backend.bases_free(h)
    assert inf == einf
    assert (got == exp).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 16])
def test_msm_window_independent(backend, curve, c):
    n = 777
    k, B = _bases(curve, n, 7)
    S = ol.random_scalars(curve, n, 8)
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    h = backend.bases_upload(curve.cid, B)
    backend.set_msm_window(c)
    try:
        got, inf = backend.msm(h, S)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)
    assert inf == einf and (got == exp).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_edge_inputs(backend, curve):
    """scalar 0 / 1 / r-1, infinity bases, repeated points, P and -P in one bucket, all scalars equal."""
    n = 300
    k, B = _bases(curve, n, 11)
    S = ol.random_scalars(curve, n, 12)
    r = curve.fr.p
    S[0] = 0
    S[1] = ol.ints_to_limbs([1], 4)[0]
    S[2] = ol.ints_to_limbs([r - 1], 4)[0]
    S[3] = ol.ints_to_limbs([1], 4)[0]
    B[4] = 0  # infinity
    B[5] = B[6]  # repeated point, same scalar -> doubling inside a bucket
    S[5] = S[6]
    # B[7] = -B[8] with equal scalars -> cancels inside a bucket
    pt = ol.limbs_to_point(curve, B[8], 0)
    B[7] = ol.points_to_limbs(curve, [po.g1_neg(curve, pt)])[0]
    S[7] = S[8]
    S[100:200] = S[100]  # many equal scalars -> one hot bucket per window
    S[200:260] = 0
    S[260:300] = ol.ints_to_limbs([1], 4)[0]
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    naive, ninf = ol.oracle_msm_g1(curve, B, S, algo=1)
    assert (exp == naive).all() and einf == ninf
    h = backend.bases_upload(curve.cid, B)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    assert inf == einf and (got == exp).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_result_infinity(backend, curve):
    k, B = _bases(curve, 2, 5)
    B[1] = B[0]
    r = curve.fr.p
    S = ol.ints_to_limbs([5, r - 5], 4)
    h = backend.bases_upload(curve.cid, B)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    assert inf == 1 and not got.any()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_bases_montgomery_upload_and_download(backend, curve):
    n = 64
    k, B = _bases(curve, n, 21)
    Bm = np.zeros_like(B)
    fid = 1 if curve.cid == 1 else 3
    ol.lib().zlo_field_to_mont(fid, ol.p64(B.reshape(-1)), ol.p64(Bm.reshape(-1)), B.size // ol.nlq(curve))
    h1 = backend.bases_upload(curve.cid, B, flags=ZL_CHECK)
    h2 = backend.bases_upload(curve.cid, Bm, flags=ZL_MONT | ZL_CHECK)
    assert (backend.bases_download(h1) == B).all()
    assert (backend.bases_download(h2) == B).all()
    S = ol.random_scalars(curve, n, 22)
    a = backend.msm(h1, S)
    b = backend.msm(h2, S)
    assert (a[0] == b[0]).all()
    backend.bases_free(h1)
    backend.bases_free(h2)
    # off-curve point is rejected when ZL_CHECK is set
    bad = B.copy()
    bad[3, 0] ^= np.uint64(1)
    with pytest.raises(BackendError) as ei:
        backend.bases_upload(curve.cid, bad, flags=ZL_CHECK)
    assert ei.value.code == -6


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_bases_generate_matches_oracle(backend, curve):
    n = 200
    k = ol.random_scalars(curve, n, 31)
    k[0] = 0
    k[1] = ol.ints_to_limbs([1], 4)[0]
    h = backend.bases_generate(curve.cid, k)
    got = backend.bases_download(h)
    backend.bases_free(h)
    exp = ol.oracle_g1_mul_gen(curve, k)
    assert (got == exp).all()


@pytest.mark.parametrize("curve,log_n", [(po.BLS12_381, 16), (po.BN254, 16), (po.BLS12_381, 20)], ids=["bls-2^16", "bn254-2^16", "bls-2^20"])
def test_msm_known_discrete_log(backend, curve, log_n):
    """Full-size exact check without a CPU MSM: P_i = k_i G  =>  MSM(s, P) = (sum s_i k_i) G."""
    n = 1 << log_n
    k = ol.random_scalars(curve, n, 41)
    S = ol.random_scalars(curve, n, 42)
    h = backend.bases_generate(curve.cid, k)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    r = curve.fr.p
    ki, si = ol.limbs_to_ints(k), ol.limbs_to_ints(S)
    dot = sum(a * b for a, b in zip(ki, si)) % r
    exp = po.g1_mul(curve, dot, po.g1_generator(curve))
    assert ol.limbs_to_point(curve, got, inf) == exp


def test_msm_skewed_scalars_bls(backend):
    """Groth16-witness-like distribution: 50% zeros, 25% ones, rest uniform (giant bucket path)."""
    curve = po.BLS12_381
    n = 1 << 16
    k = ol.random_scalars(curve, n, 51)
    S = ol.random_scalars(curve, n, 52)
    S[: n // 2] = 0
    S[n // 2: 3 * n // 4] = ol.ints_to_limbs([1], 4)[0]
    S[3 * n // 4: 7 * n // 8] = ol.ints_to_limbs([3], 4)[0]
    h = backend.bases_generate(curve.cid, k)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    r = curve.fr.p
    dot = sum(a * b for a, b in zip(ol.limbs_to_ints(k), ol.limbs_to_ints(S))) % r
    assert ol.limbs_to_point(curve, got, inf) == po.g1_mul(curve, dot, po.g1_generator(curve))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("c", [16, 17, 19, 21])
def test_msm_precomputed_table_matches_plain(backend, curve, c):
    """zl_bases_precompute (table of 2^(c w) P_i, merged bucket set, two-level sort) must not change any result."""
    n = 5000
    k, B = _bases(curve, n, 71)
    S = ol.random_scalars(curve, n, 72)
    S[0] = 0
    S[1] = ol.ints_to_limbs([1], 4)[0]
    S[2] = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]
    S[100:400] = S[100]
    B[5] = 0
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    h = backend.bases_upload(curve.cid, B)
    plain = backend.msm(h, S)
    backend.bases_precompute(h, c)
    got, inf = backend.msm(h, S)
    sub, sinf = backend.msm(h, S[1000:3000], first=1000)  # sub-range of a precomputed handle
    backend.bases_free(h)
    assert inf == einf and (got == exp).all() and (plain[0] == exp).all()
    e2, e2inf = ol.oracle_msm_g1(curve, B[1000:3000], S[1000:3000], algo=0, threads=8)
    assert sinf == e2inf and (sub == e2).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("c", [17, 18, 19, 20])
def test_msm_plain_wide_windows_match_oracle(backend, curve, c):
    """plain (no table) windows wider than 16 bits: every window keeps its own bucket set and the three-level sort runs over
    (window, bucket) ids; the hierarchical bucket reduction has 6-7 levels here.  Must not change any result."""
    n = 4000
    k, B = _bases(curve, n, 171)
    S = ol.random_scalars(curve, n, 172)
    S[0] = 0
    S[1] = ol.ints_to_limbs([1], 4)[0]
    S[2] = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]
    S[3] = ol.ints_to_limbs([(1 << (c - 1))], 4)[0]      # digit exactly at the sign boundary
    S[4] = ol.ints_to_limbs([(1 << (c - 1)) + 1], 4)[0]  # first negative digit with a carry
    S[100:300] = S[100]
    B[5] = 0
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=8)
    h = backend.bases_upload(curve.cid, B)
    backend.set_msm_window(c)
    try:
        got, inf = backend.msm(h, S)
        assert backend.last_timing().window_bits == c
        sub, sinf = backend.msm(h, S[500:1500], first=500)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)
    assert inf == einf and (got == exp).all()
    e2, e2inf = ol.oracle_msm_g1(curve, B[500:1500], S[500:1500], algo=0, threads=8)
    assert sinf == e2inf and (sub == e2).all()


def test_msm_precomputed_known_discrete_log_2_20(backend):
    curve = po.BLS12_381
    n = 1 << 20
    k = ol.random_scalars(curve, n, 81)
    S = ol.random_scalars(curve, n, 82)
    S[: n // 4] = 0
    S[n // 4: n // 2] = ol.ints_to_limbs([1], 4)[0]  # scalar-1 bypass list
    S[n // 2: 3 * n // 4] = ol.ints_to_limbs([2], 4)[0]  # one giant bucket (two-stage partial merge)
    S[3 * n // 4: 3 * n // 4 + 5000] = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]  # a big (single-block) bucket, negative digit
    h = backend.bases_generate(curve.cid, k)
    backend.bases_precompute(h, 0)
    got, inf = backend.msm(h, S)
    backend.bases_free(h)
    dot = sum(a * b for a, b in zip(ol.limbs_to_ints(k), ol.limbs_to_ints(S))) % curve.fr.p
    assert ol.limbs_to_point(curve, got, inf) == po.g1_mul(curve, dot, po.g1_generator(curve))


def _oracle_point(curve, dot: int) -> np.ndarray:
    """canonical affine x||y of dot * G from the CPU oracle (zlo_g1_mul_gen, one scalar): the expected answer of a known-discrete-log MSM
    must not come from the library under test (its device generator is compared with the oracle separately, test_bases_generate_matches_oracle)"""
    return ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([dot], 4))[0]


@pytest.mark.parametrize("mode", ["table_c22", "plain"])
def test_msm_known_discrete_log_2_24(backend, mode):
    """The exact configurations bench.py times (BASELINE metric: 2^24 BLS12-381 G1 points), checked exactly, not on a prefix:
    table_c22 = zl_bases_precompute(h, 22) (12 windows, ONE merged set of 2^21 buckets, 64 sort groups, 128-entry chunks);
    plain = no per-key work (what multi_scalar_mul(bases, scalars) is).  Expected point = (sum s_i k_i mod r) G, one O(n) dot product."""
    import torch

    from openzl_amd.selfcheck import dot_mod_r

    curve = po.BLS12_381
    n = 1 << 24
    k = ol.random_scalars(curve, n, 2401)
    S = ol.random_scalars(curve, n, 2402)
    S[5] = 0
    S[6] = ol.ints_to_limbs([1], 4)[0]
    S[7] = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]
    h = backend.bases_generate(curve.cid, k)
    if mode == "table_c22":
        backend.bases_precompute(h, 22)
    d_s = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
    # a second, different scalar vector through the pipelined batch entry point (three streams, rotating buffer sets): a cross-job
    # buffer race cannot hide behind identical inputs
    S2 = np.ascontiguousarray(S[::-1])
    d_s2 = torch.from_numpy(S2.view(np.int64)).cuda()
    torch.cuda.synchronize()
    parts = backend.msm_batch_partial_dev(h, [d_s.data_ptr(), d_s2.data_ptr(), d_s.data_ptr(), d_s2.data_ptr()], n)
    tm = backend.last_timing()
    backend.bases_free(h)
    r = curve.fr.p
    exp1 = _oracle_point(curve, dot_mod_r(S, k, r))  # one scalar multiplication on the CPU oracle: independent of the library under test
    exp2 = _oracle_point(curve, dot_mod_r(S2, k, r))
    assert not inf and (got == exp1).all()
    for j, e in enumerate((exp1, exp2, exp1, exp2)):
        xy, pinf = backend.partials_sum(curve.cid, parts[j:j + 1])
        assert not pinf and (xy == e).all(), j
    if mode == "table_c22":
        assert tm.window_bits == 22


@pytest.mark.parametrize("mode", ["plain", "table_c20"])
def test_msm_known_discrete_log_bn254_2_22(backend, mode):
    """BN254 G1 at 2^22 points (round 4: its base field moved to 10 x 28-bit lazily reduced limbs): the wide three-level sort, 64-entry chunks, the
    pipelined batch, plain and with a window table -- exact against (sum s_i k_i mod r) G from the CPU oracle's scalar multiplication."""
    import torch

    curve = po.BN254
    n, r = 1 << 22, curve.fr.p
    from openzl_amd.selfcheck import dot_mod_r

    k = ol.random_scalars(curve, n, 2254)  # uniform in [0, r): uniform points of the group
    S = ol.random_scalars(curve, n, 2255)
    S[5] = 0
    S[6] = ol.ints_to_limbs([1], 4)[0]
    S[7] = ol.ints_to_limbs([r - 1], 4)[0]
    S2 = np.ascontiguousarray(S[::-1])
    h = backend.bases_generate(curve.cid, k)
    if mode == "table_c20":
        backend.bases_precompute(h, 20)
    d1, d2 = torch.from_numpy(S.view(np.int64)).cuda(), torch.from_numpy(S2.view(np.int64)).cuda()
    torch.cuda.synchronize()
    got, inf = backend.msm_dev(h, d1.data_ptr(), n)
    parts = backend.msm_batch_partial_dev(h, [d1.data_ptr(), d2.data_ptr(), d1.data_ptr()], n)
    backend.bases_free(h)
    exp1 = _oracle_point(curve, dot_mod_r(S, k, r))
    exp2 = _oracle_point(curve, dot_mod_r(S2, k, r))
    assert not inf and (got == exp1).all()
    for j, e in enumerate((exp1, exp2, exp1)):
        xy, pinf = backend.partials_sum(curve.cid, parts[j:j + 1])
        assert not pinf and (xy == e).all(), j


@pytest.mark.parametrize("case", ["uniform", "all_equal", "two_values_and_negations"])
def test_msm_wide_path_unaligned_adversarial(backend, case):
    """The three-level sort over (window, bucket) ids on a size that is a multiple of nothing (unaligned slices, partial tiles) and on
    scalar distributions that put everything into a handful of buckets: all scalars equal (one giant bucket per window, merged in two
    stages, sorted by the tiled fine sort) and two values with their negatives (P and -P meet in the same buckets).  Exact."""
    import torch

    from openzl_amd.selfcheck import dot_mod_r

    curve = po.BLS12_381
    r = curve.fr.p
    n = (1 << 21) + 12345
    rng = np.random.Generator(np.random.PCG64(4242))
    k = ol.random_scalars(curve, n, 4241)  # uniform in [0, r): uniform points of the group
    if case == "uniform":
        S = ol.random_scalars(curve, n, 4243)
    elif case == "all_equal":
        S = np.repeat(ol.ints_to_limbs([0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % r], 4), n, axis=0)
    else:
        a, b = 0x0FEDCBA987654321 << 130 | 0x777, (1 << 200) + 12345
        vals = ol.ints_to_limbs([a, r - a, b, r - b], 4)
        S = np.ascontiguousarray(vals[rng.integers(0, 4, size=n)])
    h = backend.bases_generate(curve.cid, k)
    d_s = torch.from_numpy(np.ascontiguousarray(S).view(np.int64)).cuda()
    torch.cuda.synchronize()
    exp = _oracle_point(curve, dot_mod_r(np.ascontiguousarray(S), k, r))
    try:
        for c in (0, 19):
            backend.set_msm_window(c)
            got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
            assert not inf and (got == exp).all(), (case, c)
        backend.set_msm_window(0)
        backend.bases_precompute(h, 20)  # the merged bucket set of the table mode on the same inputs
        got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
        assert not inf and (got == exp).all(), (case, "table")
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)


def _dot_mod_r_u64k(S: np.ndarray, k64: np.ndarray, r: int) -> int:
    """sum_i S_i * k_i mod r for (n,4) u64 scalars and u64 multipliers, exact, vectorised (32-bit limb products split into halves
    so that 2^26 of them sum inside a u64)."""
    s32 = np.ascontiguousarray(S).view(np.uint32).reshape(-1, 8)
    k32 = np.ascontiguousarray(k64).view(np.uint32).reshape(-1, 2)
    total = 0
    for a in range(8):
        sa = s32[:, a].astype(np.uint64)
        for b in range(2):
            prod = sa * k32[:, b].astype(np.uint64)
            lo = int((prod & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
            hi = int((prod >> np.uint64(32)).sum(dtype=np.uint64))
            total += (lo + (hi << 32)) << (32 * (a + b))
    return total % r


def test_config4_eight_shards_of_2_23(backend):
    """BASELINE config 4 on one device: n = 2^26 as 8 contiguous shards of 2^23, each shard's points generated
    on the device from its own multipliers, one un-normalised partial per shard (what a rank all-gathers), folded by
    zl_partials_sum -- checked exactly against (sum s_i k_i) G."""
    import torch

    from openzl_amd.selfcheck import dot_mod_r

    curve = po.BLS12_381
    shards, n_s = 8, 1 << 23
    r = curve.fr.p
    parts = []
    dot = 0
    for g in range(shards):
        k = ol.random_scalars(curve, n_s, 9000 + g)  # uniform in [0, r): uniform points of the group
        S = ol.random_scalars(curve, n_s, 9100 + g)
        if g == 3:  # Groth16-witness-like shard: half zeros, a quarter ones
            S[: n_s // 2] = 0
            S[n_s // 2: 3 * n_s // 4] = ol.ints_to_limbs([1], 4)[0]
        h = backend.bases_generate(curve.cid, k)
        d_s = torch.from_numpy(S.view(np.int64)).cuda()
        torch.cuda.synchronize()
        parts.append(backend.msm_partial_dev(h, d_s.data_ptr(), n_s))
        backend.bases_free(h)
        del d_s
        dot = (dot + dot_mod_r(S, k, r)) % r
    got, inf = backend.partials_sum(curve.cid, np.stack(parts))
    assert ol.limbs_to_point(curve, got, inf) == po.g1_mul(curve, dot, po.g1_generator(curve))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("table", [False, True], ids=["plain", "table"])
def test_msm_pipelined_batch_equals_single_calls(backend, curve, table):
    """zl_msm_batch_partial_dev (three-stream pipeline over three rotating buffer sets) against one zl_msm_partial_dev per scalar vector, and the
    oracle for the first one; 5 MSMs so that buffer sets are reused."""
    import torch

    n = 6000
    k, B = _bases(curve, n, 91)
    h = backend.bases_upload(curve.cid, B)
    if table:
        backend.bases_precompute(h, 16)
    vecs = []
    for j in range(5):
        S = ol.random_scalars(curve, n, 920 + j)
        if j == 1:
            S[: n // 2] = ol.ints_to_limbs([1], 4)[0]   # scalar-1 bypass list differs per job
        if j == 3:
            S[:] = 0                                      # an all-zero job in the middle of the pipeline
        vecs.append(S)
    dev = [torch.from_numpy(S.view(np.int64)).cuda() for S in vecs]
    torch.cuda.synchronize()
    batch = backend.msm_batch_partial_dev(h, [t.data_ptr() for t in dev], n)
    for j in range(5):
        single = backend.msm_partial_dev(h, dev[j].data_ptr(), n)
        a, ai = backend.partials_sum(curve.cid, batch[j:j + 1])
        b, bi = backend.partials_sum(curve.cid, single.reshape(1, -1))
        assert ai == bi and (a == b).all(), j
    exp, einf = ol.oracle_msm_g1(curve, B, vecs[0], algo=0, threads=8)
    got, inf = backend.partials_sum(curve.cid, batch[0:1])
    backend.bases_free(h)
    assert inf == einf and (got == exp).all()


@pytest.mark.parametrize("table", [False, True], ids=["plain", "table"])
def test_msm_side_by_side_large_jobs_phase_major(backend, table):
    """Four jobs of 2^17 points in one batch take the phase-major issue order of msm_run_jobs_t (csrc/zl_msm.hip: jobs side by side on the lanes,
    at most four of them, the biggest >= 2^17 points: all sorts first, then accumulations and tails -- Groth16's four G1 MSMs of a large circuit).
    Every job against a single call and against (sum s_i k_i) G; one job is witness-like (zeros and ones), one is all zeros."""
    import torch

    curve = po.BLS12_381
    n, r = 1 << 17, curve.fr.p
    rng = np.random.Generator(np.random.PCG64(4242))
    k64 = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = k64
    h = backend.bases_generate(curve.cid, k)
    if table:
        backend.bases_precompute(h, 16)
    vecs = []
    for j in range(4):
        S = ol.random_scalars(curve, n, 4300 + j)
        if j == 1:
            S[: n // 2] = 0
            S[n // 2: 3 * n // 4] = ol.ints_to_limbs([1], 4)[0]
        if j == 2:
            S[:] = 0
        vecs.append(S)
    dev = [torch.from_numpy(S.view(np.int64)).cuda() for S in vecs]
    torch.cuda.synchronize()
    G = po.g1_generator(curve)
    for rep in range(2):  # the second batch reuses the lanes' buffers and the ctx's event pool
        batch = backend.msm_batch_partial_dev(h, [t.data_ptr() for t in dev], n)
        for j in range(4):
            single = backend.msm_partial_dev(h, dev[j].data_ptr(), n)
            a, ai = backend.partials_sum(curve.cid, batch[j:j + 1])
            b, bi = backend.partials_sum(curve.cid, single.reshape(1, -1))
            assert ai == bi and (a == b).all(), (rep, j)
            assert ol.limbs_to_point(curve, a, ai) == po.g1_mul(curve, _dot_mod_r_u64k(vecs[j], k64, r), G), (rep, j)
    backend.bases_free(h)


def test_msm_glv_edge_scalars(backend):
    """BLS12-381 G1 plain MSMs split every scalar with the endomorphism (k = k1 + k2 lambda, both halves balanced to 127 bits; csrc/zl_msm.hip
    k_glv_split).  Scalars that sit on the decision boundaries of the split -- multiples of lambda, lambda / 2 and its neighbours, 0, 1, r - 1 --
    beside random ones, every sort path (LDS c <= 15, the wide sort at c = 16 and c = 18), single call and pipelined batch, against the oracle."""
    import torch

    curve = po.BLS12_381
    r = curve.fr.p
    z = 0xD201000000010000
    lam = z * z - 1
    assert lam * lam + lam + 1 == r
    half = lam >> 1
    edge = [0, 1, 2, lam - 1, lam, lam + 1, half, half + 1, half - 1, half * lam, (half + 1) * lam, (half + 1) * lam + half + 1, (half + 1) * lam + half,
            half * lam + half, half * lam + half + 1, r - 1, r - 2, r - lam, r - lam - 1, (lam + 1) * lam % r, (half + 2) * lam - 1, 3 * lam, lam * lam % r]
    n = 3000
    S = ol.random_scalars(curve, n, 777)
    S[: len(edge)] = ol.ints_to_limbs([e % r for e in edge], 4)
    S[100:140] = ol.ints_to_limbs([(j * lam + (half if j & 1 else 0)) % r for j in range(40)], 4)
    k, B = _bases(curve, n, 778)
    h = backend.bases_upload(curve.cid, B)
    exp, einf = ol.oracle_msm_g1(curve, B, S, algo=0, threads=4)
    d_s = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    try:
        for c in (0, 7, 12, 15, 16, 18):
            backend.set_msm_window(c)
            got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
            assert inf == einf and (got == exp).all(), c
        backend.set_msm_window(0)
        parts = backend.msm_batch_partial_dev(h, [d_s.data_ptr()] * 4, n)
        for j in range(4):
            xy, pinf = backend.partials_sum(curve.cid, parts[j:j + 1])
            assert pinf == einf and (xy == exp).all(), j
        # every edge scalar alone (n = 1): the split's sign / carry cases one by one (window forced: tiny inputs would otherwise pick c <= 3,
        # where the endomorphism is not used)
        backend.set_msm_window(8)
        for e in edge:
            s1 = ol.ints_to_limbs([e % r], 4)
            got, inf = backend.msm(h, s1)
            e1, i1 = ol.oracle_msm_g1(curve, B[:1], s1)
            assert inf == i1 and (got == e1).all(), hex(e)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)


def test_msm_glv_edge_scalars_bn254(backend):
    """BN254 G1 has the same endomorphism; r is not lambda^2 + lambda + 1 with a short lambda there, so the split is the two-dimensional lattice one
    (csrc/zl_msm_endo.h k_glv_split_lattice, constants from tools/gen_bn254_glv.py).  Scalars on its decision boundaries -- multiples of lambda and of
    the basis entries and their neighbours, 0, 1, r - 1, scalars in [r, 2^254) (reduced once) -- beside random ones, every sort path, single call and
    pipelined batch, and every edge scalar alone, against the oracle."""
    import torch

    curve = po.BN254
    r = curve.fr.p
    lam = 0xB3C4D79D41A917585BFC41088D8DAAA78B17EA66B99C90DD
    assert (lam * lam + lam + 1) % r == 0
    a1, b1 = 9931322734385697763, 147946756881789319000765030803803410728  # |v1|; v2 = (b1 + a1, a1)
    a2 = a1 + b1
    assert (a1 - b1 * lam) % r == 0 and (a2 + a1 * lam) % r == 0
    edge = [0, 1, 2, lam - 1, lam, lam + 1, r - 1, r - 2, r - lam, r - lam - 1, (r - 1) // 2, (r + 1) // 2, a1, a1 - 1, a1 + 1, b1, b1 - 1, b1 + 1, a2, a2 + 1, r - a1, r - b1,
            r - a2, (b1 >> 1) * lam % r, ((b1 >> 1) + 1) * lam % r, (a1 >> 1) * lam % r, 3 * lam % r, lam * lam % r, (b1 >> 1), (b1 >> 1) + 1, (a2 >> 1), (a2 >> 1) + 1,
            ((a2 >> 1) + (a1 >> 1) * lam) % r, ((a2 >> 1) + 1 + ((a1 >> 1) + 1) * lam) % r]
    wide = [r, r + 1, r + lam % (2 ** 254 - r), 2 ** 254 - 1, 2 ** 254 - 2, r + a1, r + 12345]
    assert all(r <= x < 2 ** 254 for x in wide)
    n = 3000
    S = ol.random_scalars(curve, n, 787)
    S_red = S.copy()
    S[: len(edge)] = ol.ints_to_limbs([e % r for e in edge], 4)
    S_red[: len(edge)] = S[: len(edge)]
    S[60:60 + len(wide)] = ol.ints_to_limbs(wide, 4)
    S_red[60:60 + len(wide)] = ol.ints_to_limbs([x - r for x in wide], 4)
    mults = [(j * lam + ((b1 >> 1) if j & 1 else 0)) % r for j in range(40)]
    S[100:140] = ol.ints_to_limbs(mults, 4)
    S_red[100:140] = S[100:140]
    k, B = _bases(curve, n, 788)
    h = backend.bases_upload(curve.cid, B)
    exp, einf = ol.oracle_msm_g1(curve, B, S_red, algo=0, threads=4)
    d_s = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    try:
        for c in (0, 7, 12, 15, 16, 18):
            backend.set_msm_window(c)
            got, inf = backend.msm_dev(h, d_s.data_ptr(), n)
            assert inf == einf and (got == exp).all(), c
        backend.set_msm_window(0)
        parts = backend.msm_batch_partial_dev(h, [d_s.data_ptr()] * 4, n)
        for j in range(4):
            xy, pinf = backend.partials_sum(curve.cid, parts[j:j + 1])
            assert pinf == einf and (xy == exp).all(), j
        backend.set_msm_window(8)
        for e in edge + wide:
            s1 = ol.ints_to_limbs([e], 4)
            got, inf = backend.msm(h, s1)
            e1, i1 = ol.oracle_msm_g1(curve, B[:1], ol.ints_to_limbs([e % r], 4))
            assert inf == i1 and (got == e1).all(), hex(e)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)


def test_msm_scalars_between_r_and_2_255(backend):
    """scalars in [r, 2^255) pass the MODULUS_BITS check but are not canonical.  The plain path returns the sum mod r for them;
    the endomorphism splits (k_glv_split for G1, k_gls_split for G2; n <= 2^19) used to wrap (floor(k / lambda) > lambda + 1) and return a wrong point.
    They now reduce such a scalar once: k = r, r + 1, r + lambda, 2^255 - 1, r + random, mixed with canonical scalars, every sort path, against the
    oracle on the scalars reduced mod r."""
    import groth16_util as gu
    from openzl_amd.backend import ZL_G2

    curve = po.BLS12_381
    r = curve.fr.p
    z = 0xD201000000010000
    lam = z * z - 1
    rnd = ol.limbs_to_ints(ol.random_scalars(curve, 40, 4242))
    wide = [r, r + 1, r + lam, r + lam + 1, (1 << 255) - 1, (1 << 255) - 2, r + (lam >> 1), r + z, r + z ** 3] + [r + (x % ((1 << 255) - r)) for x in rnd]
    assert all(r <= x < (1 << 255) for x in wide)
    n = 1200
    S = ol.random_scalars(curve, n, 4243)
    S[5:5 + len(wide)] = ol.ints_to_limbs(wide, 4)
    S_red = S.copy()
    S_red[5:5 + len(wide)] = ol.ints_to_limbs([x - r for x in wide], 4)
    k, B = _bases(curve, n, 4244)
    h = backend.bases_upload(curve.cid, B)
    exp, einf = ol.oracle_msm_g1(curve, B, S_red, algo=0, threads=4)
    try:
        for c in (0, 8, 16, 18):
            backend.set_msm_window(c)
            got, inf = backend.msm(h, S)
            assert inf == einf and (got == exp).all(), c
        backend.set_msm_window(8)
        for j, x in enumerate(wide[:9]):
            got, inf = backend.msm(h, ol.ints_to_limbs([x], 4))
            e1, i1 = ol.oracle_msm_g1(curve, B[:1], ol.ints_to_limbs([x - r], 4))
            assert inf == i1 and (got == e1).all(), hex(x)
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)
    # G2 (four base-|z| digits: a scalar >= |z|^4 > r would not fit four digits)
    n2 = 300
    B2 = gu.g2_mul_gen(curve, ol.limbs_to_ints(ol.random_scalars(curve, n2, 4245)))
    S2, S2_red = S[:n2].copy(), S_red[:n2].copy()
    import ctypes as C

    exp2 = np.zeros(4 * ol.nlq(curve), dtype=np.uint64)
    einf2 = C.c_uint8(0)
    assert ol.lib().zlo_msm_g2(curve.cid, ol.p64(B2), 0, ol.p64(S2_red), n2, 0, 4, ol.p64(exp2), C.byref(einf2)) == 0
    einf2 = einf2.value
    h2 = backend.bases_upload(curve.cid, B2, group=ZL_G2)
    try:
        got_red, inf_red = backend.msm(h2, S2_red)
        for c in (0, 7, 16):
            backend.set_msm_window(c)
            got, inf = backend.msm(h2, S2)
            assert inf == inf_red and (got == got_red).all(), c
            assert inf == einf2 and (got == exp2).all(), c
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h2)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_msm_ranges_of_one_handle_and_repeated_points(backend, curve):
    """Small MSMs over different ranges of ONE handle, interleaved: the endomorphism images of the bases are kept with the handle for the first
    range used (zl_bases::d_endo, csrc/zl_msm.hip MsmJob::alloc) and every other range must fall back to its per-call copy.  The bases repeat
    (P, P, -P, ...), so the four-lane additions of the small path (csrc/zl_quad.h: accumulate, merge, level 0, tree) meet P + P, P - P and
    infinity inside buckets; all against the oracle."""
    n = 1500
    k, B = _bases(curve, n, 901)
    B = B.copy()
    B[10:40] = B[9]            # a run of equal points
    neg = ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([(curve.fr.p - int(x)) % curve.fr.p for x in ol.limbs_to_ints(k[50:60])], 4))
    B[60:70] = neg             # B[60 + j] = -B[50 + j]
    S = ol.random_scalars(curve, n, 902)
    S[10:40] = S[9]            # equal scalars on equal points: doublings inside one bucket
    S[60:70] = S[50:60]        # s P + s (-P): cancellations inside one bucket
    h = backend.bases_upload(curve.cid, B)
    try:
        for first, cnt in ((0, n), (1, n - 1), (0, n), (7, 700), (1, n - 1), (0, 64), (0, n)):
            got, inf = backend.msm(h, S[first:first + cnt], first=first)
            exp, einf = ol.oracle_msm_g1(curve, B[first:first + cnt], S[first:first + cnt], algo=0, threads=4)
            assert inf == einf and (got == exp).all(), (first, cnt)
    finally:
        backend.bases_free(h)


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("log_n,plan", [(20, "16,17,18"), (22, "19,20")], ids=["2^20-lds-sort", "2^22-wide-sort"])
@pytest.mark.parametrize("case", ["uniform", "skewed"])
def test_msm_host_scalars_in_shards_with_adversarial_contents(backend, curve, log_n, plan, case):
    """zl_msm with HOST scalars (what multi_scalar_mul is handed) cuts a large input into growing shards whose copies hide under the earlier shards' work; every shard
    is an MSM job of its own and the partials are folded.  Driven here at sizes the check finishes quickly (ZL_TUNE_HOST_CHUNK_MIN_LOG / ZL_TUNE_HOST_SHARDS: three
    resp. two leading shards + the rest, unaligned total), on uniform scalars and on a distribution built to stress single shards: a shard whose scalars are all
    zero, all equal (one giant bucket per window in that shard only), scalars 1 and r - 1 (the scalar-1 bypass), negations of the previous shard's scalars over the
    same points.  Exact against (sum s_i k_i) G.  (Round 5's carried bucket set across the shards -- exact, but slower -- was removed in round 6; this was its test.)"""
    import os

    r = curve.fr.p
    n = (1 << log_n) + 777
    rng = np.random.Generator(np.random.PCG64(99 + log_n + curve.cid))
    k64 = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    if case == "skewed":
        k64[1 << 17:(1 << 17) + 4096] = k64[0:4096]  # the same points again in the second shard ...
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = k64
    S = ol.random_scalars(curve, n, 100 + log_n)
    if case == "skewed":
        sizes = [1 << int(x) for x in plan.split(",")]
        o1, o2 = sizes[0], sizes[0] + sizes[1]
        S[0:16] = 0
        S[16:32] = ol.ints_to_limbs([1], 4)[0]
        S[32:48] = ol.ints_to_limbs([r - 1], 4)[0]
        S[o1:o1 + 4096] = ol.ints_to_limbs([(r - v) % r for v in ol.limbs_to_ints(S[0:4096])], 4)  # ... with the negated scalars: k s - k s in carried buckets
        S[o1 + 8192:o2] = 0 if len(sizes) > 2 else S[o1 + 8192:o2]                                   # (three-shard plan: most of the second shard empty)
        S[o2:o2 + 50000] = ol.ints_to_limbs([0xABCDEF0123456789ABCDEF0123456789ABCDEF0123456789 % r], 4)[0]  # a giant bucket per window in the next shard
        S[o2 + 50000:o2 + 50016] = ol.ints_to_limbs([1], 4)[0]
    S = np.ascontiguousarray(S)
    h = backend.bases_generate(curve.cid, k)
    exp = _oracle_point(curve, _dot_mod_r_u64k(S, k64, r))
    old = {kk: os.environ.get(kk) for kk in ("ZL_TUNE_HOST_CHUNK_MIN_LOG", "ZL_TUNE_HOST_SHARDS")}
    try:
        os.environ["ZL_TUNE_HOST_CHUNK_MIN_LOG"] = "18"
        os.environ["ZL_TUNE_HOST_SHARDS"] = plan
        for rep in range(2):
            got, inf = backend.msm(h, S)
            assert not inf and (got == exp).all(), rep
    finally:
        for kk, v in old.items():
            if v is None:
                os.environ.pop(kk, None)
            else:
                os.environ[kk] = v
        backend.bases_free(h)
