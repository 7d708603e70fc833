"""GPU parity of the distributed NTT legs (zl_ntt_cross_dev + zl_ntt_dev with one-sided Montgomery flags).

A 1-GPU box runs the G ranks as virtual ranks on one device: every rank's slice goes through the same kernels a real rank
launches and the all-to-all is emulated by a tensor shuffle ("validate G-way logic with virtual shards").
The result must be bit-identical to the CPU oracle's transform of the whole vector (and, at 2^22, to the single-device zl_ntt)."""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd.backend import ZL_COSET, ZL_INVERSE, ZL_MONT_IN, ZL_MONT_OUT
from openzl_amd.sharded import block_column_slice, cyclic_slice

pytestmark = pytest.mark.gpu
CURVES = [po.BLS12_381, po.BN254]


def _virtual_transform(backend, curve, slices, log_n, log_g, inverse, coset):
    """slices: per-rank numpy (M, 4) uint64 canonical -> per-rank numpy result, emulating the exchange on one device."""
    import torch

    G = 1 << log_g
    M = 1 << (log_n - log_g)
    B = M // G
    dev = [torch.from_numpy(s.view(np.int64).copy()).cuda() for s in slices]
    torch.cuda.synchronize()
    base = (ZL_INVERSE if inverse else 0) | (ZL_COSET if coset else 0)
    plain = ZL_INVERSE if inverse else 0
    for g in range(G):
        if not inverse:
            backend.ntt_cross_dev(curve.cid, dev[g].data_ptr(), log_n, log_g, g, base | ZL_MONT_OUT)
        else:
            backend.ntt_dev_flags(curve.cid, dev[g].data_ptr(), log_n - log_g, plain | ZL_MONT_OUT)
    backend.sync()
    # all_to_all_single: rank d receives chunk d of every rank g, stored at position g
    recv = [torch.cat([dev[g].view(G, B, 4)[d] for g in range(G)]).contiguous() for d in range(G)]
    torch.cuda.synchronize()
    for d in range(G):
        if not inverse:
            backend.ntt_dev_flags(curve.cid, recv[d].data_ptr(), log_n - log_g, plain | ZL_MONT_IN)
        else:
            backend.ntt_cross_dev(curve.cid, recv[d].data_ptr(), log_n, log_g, d, base | ZL_MONT_IN)
    backend.sync()
    return [t.cpu().numpy().view(np.uint64) for t in recv]


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_g", [1, 2, 3, 4])
@pytest.mark.parametrize("coset", [False, True])
def test_virtual_ranks_match_oracle(backend, curve, log_g, coset):
    for log_n in (2 * log_g, 2 * log_g + 1, 12, 15):
        x = ol.random_scalars(curve, 1 << log_n, 300 + log_n)
        X = ol.oracle_ntt(curve, x, inverse=False, coset=coset)
        G = 1 << log_g
        got = _virtual_transform(backend, curve, [block_column_slice(x, log_g, g) for g in range(G)], log_n, log_g, False, coset)
        for g in range(G):
            assert (got[g] == cyclic_slice(X, log_g, g)).all(), (log_n, g)
        back = _virtual_transform(backend, curve, got, log_n, log_g, True, coset)
        for g in range(G):
            assert (back[g] == block_column_slice(x, log_g, g)).all(), (log_n, g)


def test_virtual_ranks_match_single_device_2_22(backend):
    """8 virtual ranks x 2^19 elements vs one zl_ntt of 2^22 (itself oracle-checked up to 2^18 and by properties at 2^24)."""
    curve, log_n, log_g = po.BLS12_381, 22, 3
    x = ol.random_scalars(curve, 1 << log_n, 9)
    X = backend.ntt(curve.cid, x, coset=True)
    got = _virtual_transform(backend, curve, [block_column_slice(x, log_g, g) for g in range(8)], log_n, log_g, False, True)
    for g in range(8):
        assert (got[g] == cyclic_slice(X, log_g, g)).all()


def test_cross_rejects_bad_arguments(backend):
    import torch
    from openzl_amd.backend import BackendError

    t = torch.zeros(64, 4, dtype=torch.int64).cuda()
    for log_n, log_g, rank in [(6, 0, 0), (6, 5, 0), (3, 2, 0), (6, 2, 4), (40, 2, 0)]:
        with pytest.raises(BackendError):
            backend.ntt_cross_dev(po.BLS12_381.cid, t.data_ptr(), log_n, log_g, rank, 0)
    with pytest.raises(BackendError):
        backend.ntt_cross_dev(po.BLS12_381.cid, t.data_ptr(), 6, 1, 0, 1 << 10)
