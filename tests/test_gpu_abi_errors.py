"""Error behaviour of the C ABI on a live ctx: every failure is a return code, nothing aborts (the reference collapses every
upstream failure into an opaque Error, plugins/arkworks/src/groth16.rs:438-465; the codes here are the extra diagnostics)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd import BackendError, ZL_G2

pytestmark = pytest.mark.gpu
EINVAL, EHANDLE = -1, -5


def test_handle_and_range_errors(backend):
    curve = po.BLS12_381
    k = ol.random_scalars(curve, 16, 1)
    S = ol.random_scalars(curve, 16, 2)
    h = backend.bases_generate(curve.cid, k)
    with pytest.raises(BackendError) as e:
        backend.msm(h, np.concatenate([S, S]))  # more scalars than bases
    assert e.value.code == EINVAL
    with pytest.raises(BackendError) as e:
        backend.msm(h, S, first=8)  # range runs past the end
    assert e.value.code == EINVAL
    got = backend.msm(h, S[:8], first=8)  # sub-range is fine
    exp = ol.oracle_msm_g1(curve, ol.oracle_g1_mul_gen(curve, k)[8:], S[:8])
    assert (got[0] == exp[0]).all()
    backend.bases_free(h)
    backend._bases[h] = (curve.cid, 1, 16)  # pretend the handle is still alive
    with pytest.raises(BackendError) as e:
        backend.msm(h, S)
    assert e.value.code == EHANDLE
    with pytest.raises(BackendError) as e:
        backend.bases_free(h)
    assert e.value.code == EHANDLE
    assert backend.L.zl_bases_precompute(backend._ctx, 123456, 0) == EHANDLE
    assert backend.L.zl_ctx_set_msm_window(backend._ctx, 1) == EINVAL
    assert backend.L.zl_ctx_set_msm_window(backend._ctx, 99) == EINVAL


def test_ntt_argument_errors(backend):
    L = backend.L
    buf = np.zeros((4, 4), dtype=np.uint64)
    assert L.zl_ntt(backend._ctx, 7, ol.p64(buf), 2, 0) == EINVAL       # unknown curve
    assert L.zl_ntt(backend._ctx, po.BN254.cid, ol.p64(buf), 29, 0) == EINVAL  # log_n above BN254's two-adicity (28)
    assert L.zl_ntt(backend._ctx, po.BLS12_381.cid, None, 2, 0) == EINVAL
    assert L.zl_ntt_dev(backend._ctx, po.BLS12_381.cid, C.c_void_p(1), 2, 64) == EINVAL  # unknown flag bit


def test_groth16_rejects_mismatched_handles(backend):
    curve = po.BLS12_381
    import groth16_util as gu

    cs = po.poseidon_chain_circuit(curve.fr, 1)
    td = po.Groth16Trapdoor(3, 5, 7, 11, 13)
    pk = gu.setup_with_trapdoor(curve, cs, td)
    dpk = gu.upload_pk(backend, curve, pk)
    z = ol.ints_to_limbs(cs.assignment(), 4)
    r = ol.ints_to_limbs([1], 4)[0]
    try:
        bad = dict(dpk)
        bad["b_g2_query"] = dpk["a_query"]  # a G1 handle where the G2 query is expected
        with pytest.raises(BackendError) as e:
            backend.groth16_prove(curve.cid, bad, gu.r1cs_arrays(cs), z, r, r)
        assert e.value.code == EHANDLE
        short = dict(dpk)
        short["h_query"] = backend.bases_upload(curve.cid, pk["h_query"][:10])  # too few points for the domain
        with pytest.raises(BackendError) as e:
            backend.groth16_prove(curve.cid, short, gu.r1cs_arrays(cs), z, r, r)
        assert e.value.code == EINVAL
        backend.bases_free(short["h_query"])
    finally:
        gu.free_pk(backend, dpk)


def test_batch_edge_cases(backend):
    """zl_msm_batch_partial_dev: empty batch, empty MSMs, a single job, bad handle / range / null scalars -- return codes only."""
    import torch

    curve = po.BLS12_381
    n = 300
    k = ol.random_scalars(curve, n, 3)
    S = ol.random_scalars(curve, n, 4)
    h = backend.bases_generate(curve.cid, k)
    d = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert backend.msm_batch_partial_dev(h, [], n).shape == (0, 64)                       # count = 0
    empty = backend.msm_batch_partial_dev(h, [d.data_ptr()] * 3, 0)                         # n = 0: three points at infinity
    for j in range(3):
        assert backend.partials_sum(curve.cid, empty[j:j + 1])[1] == 1
    one = backend.msm_batch_partial_dev(h, [d.data_ptr()], n)                               # a batch of one = a single call
    exp = ol.oracle_msm_g1(curve, ol.oracle_g1_mul_gen(curve, k), S)
    got = backend.partials_sum(curve.cid, one)
    assert got[1] == exp[1] and (got[0] == exp[0]).all()
    sub = backend.msm_batch_partial_dev(h, [d.data_ptr()] * 4, 100, first=50)               # sub-range, four jobs
    e2 = ol.oracle_msm_g1(curve, ol.oracle_g1_mul_gen(curve, k)[50:150], S[:100])
    for j in range(4):
        g2 = backend.partials_sum(curve.cid, sub[j:j + 1])
        assert g2[1] == e2[1] and (g2[0] == e2[0]).all()
    with pytest.raises(BackendError) as e:
        backend.msm_batch_partial_dev(h, [d.data_ptr()] * 2, n, first=10)                   # range runs past the end
    assert e.value.code == EINVAL
    with pytest.raises(BackendError) as e:
        backend.msm_batch_partial_dev(h, [d.data_ptr(), 0], n)                              # a null scalar vector
    assert e.value.code == EINVAL
    with pytest.raises(BackendError) as e:
        backend.msm_batch_partial_dev(987654, [d.data_ptr()], n)
    assert e.value.code == EHANDLE
    backend.bases_free(h)


def test_groth16_rejects_malformed_csr(backend):
    """zl_r1cs_upload validates the caller's CSR once (monotone row_ptr starting at 0, column indices < n_instance + n_witness): a
    malformed matrix is ZL_EINVAL, never an out-of-bounds read in the sparse mat-vec kernel."""
    curve = po.BLS12_381
    import groth16_util as gu

    cs = po.poseidon_chain_circuit(curve.fr, 1)
    td = po.Groth16Trapdoor(3, 5, 7, 11, 13)
    pk = gu.setup_with_trapdoor(curve, cs, td)
    dpk = gu.upload_pk(backend, curve, pk)
    z = ol.ints_to_limbs(cs.assignment(), 4)
    r = ol.ints_to_limbs([1], 4)[0]
    try:
        good = gu.r1cs_arrays(cs)
        backend.groth16_prove(curve.cid, dpk, good, z, r, r)  # sanity: the well-formed system proves
        nv = good["n_instance"] + good["n_witness"]
        for key, mutate in (("A", "col"), ("B", "ptr_nonmonotone"), ("C", "ptr_start")):
            bad = dict(good)
            ptr, col, val = (a.copy() for a in good[key])
            if mutate == "col":
                col[len(col) // 2] = nv  # one past the last variable
            elif mutate == "ptr_nonmonotone":
                ptr[3] = ptr[4] + 1
            else:
                ptr[0] = 1
            bad[key] = (ptr, col, val)
            with pytest.raises(BackendError) as e:
                backend.groth16_prove(curve.cid, dpk, bad, z, r, r)
            assert e.value.code == EINVAL, mutate
    finally:
        gu.free_pk(backend, dpk)


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("n,c,table", [(1000, 0, False), (1000, 18, False), (1 << 22, 0, False), (70000, 0, True)],
                         ids=["lds_sort", "wide_c18", "2^22_spread_top_window", "table"])
def test_msm_rejects_scalars_wider_than_the_field(backend, curve, n, c, table):
    """The ABI takes canonical scalars (< r).  A scalar with bits at or above MODULUS_BITS would be cut or mis-bucketed by the window layout
    (the spread top window of the plain 17..20-bit path at n >= 2^22 ORs the point index above the digit), so every sort path flags it
    and the call fails with ZL_EINVAL instead of returning a wrong point; the same vector with the offending scalar reduced works."""
    import torch

    if curve is po.BN254 and n > 100000:
        pytest.skip("one large case is enough")
    rng = np.random.Generator(np.random.PCG64(5150 + n))
    k = np.zeros((n, 4), dtype=np.uint64)
    k[:, 0] = rng.integers(1, 1 << 62, size=n, dtype=np.uint64)
    S = ol.random_scalars(curve, n, 5151)
    h = backend.bases_generate(curve.cid, k)
    try:
        if table:
            backend.bases_precompute(h, 16)
        backend.set_msm_window(c)
        good = backend.msm(h, S)
        bad = S.copy()
        bad[n // 3, 3] |= np.uint64(1) << np.uint64(curve.fr.bits - 192)  # bit MODULUS_BITS of one scalar
        with pytest.raises(BackendError) as e:
            backend.msm(h, bad)
        assert e.value.code == EINVAL
        d_bad = torch.from_numpy(bad.view(np.int64)).cuda()
        d_ok = torch.from_numpy(S.view(np.int64)).cuda()
        torch.cuda.synchronize()
        with pytest.raises(BackendError) as e:
            backend.msm_batch_partial_dev(h, [d_ok.data_ptr(), d_bad.data_ptr(), d_ok.data_ptr()], n)
        assert e.value.code == EINVAL
        again = backend.msm(h, S)  # the ctx is usable afterwards and the flag does not stick
        assert again[1] == good[1] and (again[0] == good[0]).all()
    finally:
        backend.set_msm_window(0)
        backend.bases_free(h)
