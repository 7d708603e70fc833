"""The multi-GPU entry points of the C ABI (zl_ctx_create_multi / zl_msm_sharded / zl_ntt_sharded, include/zl_backend.h) with G virtual
ranks on ONE device: the shard -> local MSM -> all-gather -> fold and cross -> all-to-all -> local NTT paths run exactly as on G
devices, only the exchange is a device-to-device copy instead of RCCL (RCCL refuses duplicate devices; a 1-GPU box cannot test it).
Bit-exact against the oracle's single-device answers."""
import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import MultiBackend
from openzl_amd.sharded import block_column_slice, cyclic_slice

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _torch_first(backend):
    """torch must initialise HIP before the library's own contexts (see conftest.backend)"""
    yield


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_msm_sharded_virtual_ranks_match_oracle(curve, G):
    import torch

    n_s = 3000
    mb = MultiBackend([0] * G)
    try:
        assert mb.size == G and not mb.uses_rccl
        ks, Ss, hs, ds = [], [], [], []
        for g in range(G):
            k = ol.random_scalars(curve, n_s, 7000 + g)
            S = ol.random_scalars(curve, n_s - g, 7100 + g)  # ragged shards
            if g == 1:
                S[: n_s // 2] = 0
                S[n_s // 2: n_s // 2 + 50] = ol.ints_to_limbs([1], 4)[0]
            ks.append(k)
            Ss.append(S)
            hs.append(mb.ranks[g].bases_generate(curve.cid, k))
            ds.append(torch.from_numpy(S.view(np.int64)).cuda())
        torch.cuda.synchronize()
        got, inf = mb.msm_sharded(hs, [d.data_ptr() for d in ds], [S.shape[0] for S in Ss])
        dot = 0
        for k, S in zip(ks, Ss):
            dot += sum(a * b for a, b in zip(ol.limbs_to_ints(k[: S.shape[0]]), ol.limbs_to_ints(S)))
        exp = po.g1_mul(curve, dot % curve.fr.p, po.g1_generator(curve))
        assert ol.limbs_to_point(curve, got, inf) == exp
        # all shards empty -> infinity
        got, inf = mb.msm_sharded(hs, [d.data_ptr() for d in ds], [0] * G)
        assert inf == 1
        for g in range(G):
            mb.ranks[g].bases_free(hs[g])
    finally:
        mb.close()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("log_g", [1, 2, 3])
@pytest.mark.parametrize("coset", [False, True], ids=["plain", "coset"])
def test_ntt_sharded_virtual_ranks_match_oracle(curve, log_g, coset):
    import torch

    log_n = 12
    G = 1 << log_g
    x = ol.random_scalars(curve, 1 << log_n, 7200 + log_g)
    X = ol.oracle_ntt(curve, x, coset=coset)
    mb = MultiBackend([0] * G)
    try:
        # forward: block-column slices of the coefficients in, cyclic slices of the evaluations out
        ts = [torch.from_numpy(block_column_slice(x, log_g, g).view(np.int64)).cuda() for g in range(G)]
        torch.cuda.synchronize()
        mb.ntt_sharded(curve.cid, [t.data_ptr() for t in ts], log_n, coset=coset)
        for g in range(G):
            assert (ts[g].cpu().numpy().view(np.uint64) == cyclic_slice(X, log_g, g)).all(), ("forward", g)
        # inverse: back to the block-column coefficients
        mb.ntt_sharded(curve.cid, [t.data_ptr() for t in ts], log_n, inverse=True, coset=coset)
        for g in range(G):
            assert (ts[g].cpu().numpy().view(np.uint64) == block_column_slice(x, log_g, g)).all(), ("inverse", g)
    finally:
        mb.close()


def test_multi_argument_errors():
    from openzl_amd.backend import load_library
    import ctypes as C

    L = load_library()
    m = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert L.zl_ctx_create_multi(C.byref(m), ids, 0) == -1
    assert L.zl_ctx_create_multi(None, ids, 2) == -1
    assert L.zl_ctx_create_multi(C.byref(m), (C.c_int * 2)(0, 99), 2) != 0
    mb = MultiBackend([0, 0, 0])  # 3 ranks: not a power of two -> the NTT refuses, the MSM does not care
    try:
        with pytest.raises(Exception):
            mb.ntt_sharded(po.BLS12_381.cid, [1, 1, 1], 12)
    finally:
        mb.close()


def test_msm_sharded_single_rank_through_rccl(monkeypatch):
    """The RCCL branch of zl_msm_sharded (dlopen of librccl, ncclCommInitAll, grouped ncclAllGather of the partial sums) with a ONE-rank
    communicator: the only form of it a 1-GPU box can run.  Multi-rank RCCL runs only on the driver's multi-GPU node."""
    import torch

    monkeypatch.setenv("ZL_FORCE_RCCL", "1")
    curve = po.BLS12_381
    n = 2000
    mb = MultiBackend([0])
    try:
        assert mb.size == 1 and mb.uses_rccl
        k = ol.random_scalars(curve, n, 7300)
        S = ol.random_scalars(curve, n, 7301)
        h = mb.ranks[0].bases_generate(curve.cid, k)
        d = torch.from_numpy(S.view(np.int64)).cuda()
        torch.cuda.synchronize()
        got, inf = mb.msm_sharded([h], [d.data_ptr()], [n])
        dot = sum(a * b for a, b in zip(ol.limbs_to_ints(k), ol.limbs_to_ints(S))) % curve.fr.p
        assert ol.limbs_to_point(curve, got, inf) == po.g1_mul(curve, dot, po.g1_generator(curve))
        mb.ranks[0].bases_free(h)
    finally:
        mb.close()


@pytest.mark.parametrize("curve,k,G,cuts", [(po.BLS12_381, 8, 2, None), (po.BLS12_381, 8, 4, None), (po.BN254, 8, 4, None), (po.BLS12_381, 1, 4, (0.0, 0.1, 0.1)),
                                             (po.BLS12_381, 64, 3, None)], ids=["bls-k8-G2", "bls-k8-G4", "bn254-k8-G4", "bls-k1-G4-empty-slices", "bls-k64-G3"])
def test_groth16_one_proof_over_virtual_ranks(curve, k, G, cuts):
    """ONE proof over the G ranks of an mctx (zl_groth16_prove_sharded): every rank holds a contiguous slice of each of the five
    queries, rank 0 runs the witness map, every rank its five partial MSMs, the host folds and assembles -- the proof must equal the CPU oracle's (and hence
    the single-device prover's) for the same (r, s), byte for byte, for equal cuts, for three ranks (slices that are no power of two) and for cuts that leave a
    rank with EMPTY slices (rank 0 with none of the points: it still runs the witness map).  Virtual ranks on GPU 0: RCCL plays no part here, the exchange of a
    proof is the device-to-device copies of the z / h slices and 5 x 512 B of partials per rank."""
    import test_groth16 as tg

    cs, pk, arrays, z, r, s = tg._case(curve, k)
    mb = MultiBackend([0] * G)
    try:
        got = mb.groth16_prove_sharded(curve.cid, pk, arrays, z, r, s, cuts=cuts)
    finally:
        mb.close()
    exp, _ = gu.oracle_prove(curve, arrays, z, pk, r, s, threads=8)
    for g_, e in zip(got, exp):
        assert np.array_equal(np.asarray(g_), np.asarray(e))
