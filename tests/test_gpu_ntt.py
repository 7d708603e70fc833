"""GPU parity: zl_ntt (HIP multi-pass NTT) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po

pytestmark = pytest.mark.gpu
CURVES = [po.BLS12_381, po.BN254]


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 16, 17, 18])
@pytest.mark.parametrize("inverse,coset", [(False, False), (True, False), (False, True), (True, True)])
def test_ntt_matches_oracle(backend, curve, log_n, inverse, coset):
    x = ol.random_scalars(curve, 1 << log_n, 1000 + log_n)
    got = backend.ntt(curve.cid, x, inverse=inverse, coset=coset, mont=False)
    exp = ol.oracle_ntt(curve, x, inverse=inverse, coset=coset, mont=False)
    assert (got == exp).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_ntt_small_golden(backend, curve):
    """definition-level check (O(n^2) DFT in big ints) on a tiny input, all four variants"""
    xs = [1, 2, 3, 4, 5, 6, 7, 8]
    for inverse in (False, True):
        for coset in (False, True):
            got = ol.limbs_to_ints(backend.ntt(curve.cid, ol.ints_to_limbs(xs, 4), inverse=inverse, coset=coset))
            assert got == po.dft_naive(curve, xs, inverse, coset)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_ntt_montgomery_io(backend, curve):
    x = ol.random_scalars(curve, 1 << 12, 5)
    fid = 2 if curve.cid == 1 else 4
    xm = np.zeros_like(x)
    ol.lib().zlo_field_to_mont(fid, ol.p64(x.reshape(-1)), ol.p64(xm.reshape(-1)), x.shape[0])
    got_m = backend.ntt(curve.cid, xm, mont=True)
    got = np.zeros_like(x)
    ol.lib().zlo_field_from_mont(fid, ol.p64(got_m.reshape(-1)), ol.p64(got.reshape(-1)), x.shape[0])
    assert (got == ol.oracle_ntt(curve, x)).all()


@pytest.mark.parametrize("log_n", [20, 24])
def test_ntt_roundtrip_and_spot_checks_full_size(backend, log_n):
    """size-independent properties at BASELINE sizes: iNTT(NTT(x)) == x bit-exact, plus Horner spot checks
    X_k = x(w^k) for random k."""
    curve = po.BLS12_381
    n = 1 << log_n
    x = ol.random_scalars(curve, n, 77)
    X = backend.ntt(curve.cid, x)
    back = backend.ntt(curve.cid, X, inverse=True)
    assert (back == x).all()
    r = curve.fr.p
    w = po.domain_root(curve, log_n)
    m = 1 << 10  # Horner over a zero-padded low-degree input keeps the big-int check O(2^10) per point
    x2 = x.copy()
    x2[m:] = 0
    X2 = backend.ntt(curve.cid, x2)
    coeffs = ol.limbs_to_ints(x2[:m])
    rng = np.random.default_rng(3)
    for k in [0, 1, n - 1] + [int(v) for v in rng.integers(0, n, 5)]:
        wk = pow(w, k, r)
        acc = 0
        for cf in reversed(coeffs):
            acc = (acc * wk + cf) % r
        assert ol.limbs_to_ints(X2[k:k + 1])[0] == acc


@pytest.mark.parametrize("coset", [False, True], ids=["plain", "coset"])
def test_ntt_four_pass_sizes_roundtrip_and_spot_checks(backend, coset):
    """sizes n >= 25 take FOUR passes through the lazily reduced kernel (k_ntt_pass28): a second middle pass with its own row table (d_row[2]) that
    runs in place over the 40-byte scratch elements -- a path no smaller size reaches.  2^25, forward and inverse, plain and coset: the round trip is the
    identity bit for bit, and the transform of a low-degree input equals Horner's evaluation at w^k (resp. g w^k) in Python integers."""
    curve = po.BLS12_381
    log_n = 25
    n = 1 << log_n
    x = ol.random_scalars(curve, n, 78)
    X = backend.ntt(curve.cid, x, coset=coset)
    assert not (X[:8] == x[:8]).all()
    back = backend.ntt(curve.cid, X, inverse=True, coset=coset)
    assert (back == x).all()
    del X, back
    r = curve.fr.p
    w = po.domain_root(curve, log_n)
    g = 7 if coset else 1  # Fr multiplicative generator of BLS12-381 (include/zl_backend.h: ZL_COSET)
    m = 1 << 10
    x2 = np.zeros_like(x)
    x2[:m] = x[:m]
    X2 = backend.ntt(curve.cid, x2, coset=coset)
    coeffs = ol.limbs_to_ints(x2[:m])
    rng = np.random.default_rng(4)
    for k in [0, 1, n - 1, n // 2 + 1] + [int(v) for v in rng.integers(0, n, 6)]:
        pt = g * pow(w, k, r) % r
        acc = 0
        for cf in reversed(coeffs):
            acc = (acc * pt + cf) % r
        assert ol.limbs_to_ints(X2[k:k + 1])[0] == acc, k


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [1, 8, 11, 14, 17, 20])  # one, two and three passes (17: a middle pass with its row table)
def test_ntt_batch_equals_single_transforms(backend, curve, log_n):
    """zl_ntt_batch_dev (Groth16's witness map: a, b, c through each transform in one launch per pass) = the same transforms one at a time, for every
    variant, with a stride larger than the vectors (the gaps must stay untouched)"""
    import torch
    from openzl_amd.backend import ZL_COSET, ZL_INVERSE, ZL_MONT

    n = 1 << log_n
    count, stride = 3, n + 8
    x = ol.random_scalars(curve, count * stride, 4000 + log_n)
    for flags in (ZL_MONT, ZL_MONT | ZL_INVERSE, ZL_MONT | ZL_COSET, ZL_MONT | ZL_INVERSE | ZL_COSET, 0):
        d_b = torch.from_numpy(x.view(np.int64).copy()).cuda()
        d_s = torch.from_numpy(x.view(np.int64).copy()).cuda()
        backend.ntt_batch_dev(curve.cid, d_b.data_ptr(), log_n, flags, count, stride)
        for v in range(count):
            backend.ntt_dev_flags(curve.cid, d_s.data_ptr() + v * stride * 32, log_n, flags)
        torch.cuda.synchronize()
        assert torch.equal(d_b, d_s), flags
        gaps = d_b.cpu().numpy().view(np.uint64).reshape(count, stride, 4)[:, n:, :]
        assert (gaps == x.reshape(count, stride, 4)[:, n:, :]).all()
