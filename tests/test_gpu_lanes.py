"""Prover lanes (zl_ctx_fork): N host threads, one forked ctx each, over ONE device-resident proving key -- the counterpart of N threads sharing the
reference's `&ProvingContext` (plugins/arkworks/src/groth16.rs:445-457 takes the context by shared reference; the reference's ProvingContext is Send + Sync).
Every proof from every lane must be the parent's proof byte for byte (a proof is a function of key, witness and rng only)."""
import ctypes as C
import threading

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po

from openzl_amd import BackendError
from openzl_amd.backend import Circuit, Groth16Keys

pytestmark = pytest.mark.gpu
CURVES = [po.BLS12_381, po.BN254]


def _same(p, q):
    return all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p, q))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("k", [1, 24])
def test_lanes_prove_side_by_side_over_one_key(backend, curve, k):
    """three threads (the parent ctx and two lanes), 8 proofs each with two alternating witnesses, all at once: every proof equals the one the parent made alone"""
    full = Circuit(curve.cid, k)
    other = Circuit(curve.cid, k, x0=5, x1=9, witness_only=True)
    keys = Groth16Keys(backend, full, seed=31)
    lanes = [backend.fork(), backend.fork()]
    try:
        ref = [keys.prove(seed=40)[0], keys.prove(seed=41, circuit=other)[0]]
        assert keys.verify(ref[1], other.arrays()["assignment"][1:2])
        for ln in lanes:  # a lane alone
            assert _same(ref[0], keys.prove(seed=40, lane=ln)[0]) and _same(ref[1], keys.prove(seed=41, circuit=other, lane=ln)[0])
        bad, errs = [], []

        def run(ln, tid):
            try:
                for i in range(8):
                    j = (i + tid) & 1
                    p = keys.prove(seed=40 + j, circuit=other if j else None, lane=ln)[0]
                    if not _same(ref[j], p):
                        bad.append((tid, i))
            except Exception as e:  # noqa: BLE001
                errs.append(repr(e))

        for rep in range(3):
            th = [threading.Thread(target=run, args=(ln, t)) for t, ln in enumerate([None] + lanes)]
            [t.start() for t in th]
            [t.join() for t in th]
        assert not errs and not bad, (errs, bad)
    finally:
        for ln in lanes:
            ln.close()
        keys.close()
        full.close()
        other.close()


def test_lane_reads_parent_bases_and_keeps_its_own(backend):
    """an MSM on a lane over a handle of the parent (no GLV cache yet: the lane builds the shared one), the same on the parent afterwards; a lane's own upload
    gets a handle the parent does not know; lifetime rules"""
    import torch

    curve = po.BLS12_381
    n = 5000
    k = ol.random_scalars(curve, n, 777)
    S = ol.random_scalars(curve, n, 778)
    h = backend.bases_generate(curve.cid, k)
    d = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    lane = backend.fork()
    try:
        got_l, inf_l = lane.msm_dev(h, d.data_ptr(), n)
        got_p, inf_p = backend.msm_dev(h, d.data_ptr(), n)
        from openzl_amd.selfcheck import dot_mod_r

        exp = ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([dot_mod_r(S, k, curve.fr.p)], 4))[0]  # (sum s_i k_i) G from the CPU oracle
        assert not inf_l and not inf_p and (got_l == exp).all() and (got_p == exp).all()
        assert (lane.bases_download(h, 0, 3) == backend.bases_download(h, 0, 3)).all()
        h2 = lane.bases_generate(curve.cid, k[:100])
        assert h2 != h and h2 >> 48
        part = np.zeros(64, dtype=np.uint64)
        assert backend.L.zl_msm_partial_dev(backend._ctx, h2, 0, d.data_ptr(), 100, part.ctypes.data_as(C.POINTER(C.c_uint64))) == -5  # ZL_EHANDLE: the parent does not see a lane's objects
        got2, _ = lane.msm_dev(h2, d.data_ptr(), 100)
        got3, _ = backend.msm_dev(h, d.data_ptr(), 100)
        assert (got2 == got3).all()
        with pytest.raises(BackendError):
            lane.fork()                                  # one level
        with pytest.raises(BackendError):
            backend.bases_free(h)                        # a lane may be reading it
        with pytest.raises(BackendError):
            backend.bases_precompute(h, 16)
        lane.bases_free(h2)
    finally:
        lane.close()
    backend.bases_precompute(h, 16)
    got_t, _ = backend.msm_dev(h, d.data_ptr(), n)
    assert (got_t == got_p).all()
    backend.bases_free(h)


def test_decoded_context_binds_on_its_own_ctx_then_serves_lanes(backend):
    curve = po.BLS12_381
    full = Circuit(curve.cid, 2)
    wit = Circuit(curve.cid, 2, witness_only=True)
    keys = Groth16Keys(backend, full, seed=3)
    dec = Groth16Keys.from_bytes(backend, full, keys.to_bytes())
    lane = backend.fork()
    try:
        ref = keys.prove(seed=9)[0]
        with pytest.raises(BackendError):
            dec.prove(seed=9, lane=lane)          # unbound: the first proof uploads the matrices, on the context's own ctx
        assert _same(ref, dec.prove(seed=9)[0])
        assert _same(ref, dec.prove(seed=9, lane=lane)[0]) and _same(ref, dec.prove(seed=9, circuit=wit, lane=lane)[0])
        stranger = type(backend)(0)
        try:
            with pytest.raises(BackendError):
                keys.prove(seed=9, lane=stranger)  # not a lane of the keys' ctx
        finally:
            stranger.close()
    finally:
        lane.close()
        dec.close()
        keys.close()
        full.close()
        wit.close()


def test_parent_destroyed_before_its_lane_orphans_it(backend):
    """a caller error the library must survive: the lane outlives its parent -> its lookups of the parent's handles fail with ZL_EHANDLE (no dangling pointer),
    its own objects keep working, destroying it afterwards is clean"""
    import torch

    L = backend.L
    parent, lane = C.c_void_p(), C.c_void_p()
    assert L.zl_ctx_create(C.byref(parent), 0) == 0 and L.zl_ctx_fork(parent, C.byref(lane)) == 0
    curve = po.BLS12_381
    k = ol.random_scalars(curve, 64, 5)
    S = ol.random_scalars(curve, 64, 6)
    d = torch.from_numpy(S.view(np.int64)).cuda()
    torch.cuda.synchronize()
    u64p = C.POINTER(C.c_uint64)
    h, h2 = C.c_uint64(), C.c_uint64()
    assert L.zl_bases_generate(parent, curve.cid, 1, k.ctypes.data_as(u64p), 64, C.byref(h)) == 0
    assert L.zl_bases_generate(lane, curve.cid, 1, k.ctypes.data_as(u64p), 64, C.byref(h2)) == 0
    part, part2 = np.zeros(64, dtype=np.uint64), np.zeros(64, dtype=np.uint64)
    assert L.zl_msm_partial_dev(lane, h.value, 0, d.data_ptr(), 64, part.ctypes.data_as(u64p)) == 0
    L.zl_ctx_destroy(parent)
    assert L.zl_msm_partial_dev(lane, h.value, 0, d.data_ptr(), 64, part2.ctypes.data_as(u64p)) == -5
    assert L.zl_msm_partial_dev(lane, h2.value, 0, d.data_ptr(), 64, part2.ctypes.data_as(u64p)) == 0
    a, _ = backend.partials_sum(curve.cid, part.reshape(1, -1))
    b, _ = backend.partials_sum(curve.cid, part2.reshape(1, -1))
    assert (a == b).all()
    L.zl_ctx_destroy(lane)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_prove_many_is_a_stream_of_single_proofs(backend, curve):
    """zl_groth16_prove_circuits (two host threads inside the library, the ctx and a fork it keeps): proofs[i] == prove(seeds[i], circuits[i]) byte for byte;
    the error of a bad element comes back; the kept lane does NOT pin the key once the call has returned"""
    k = 6
    full = Circuit(curve.cid, k)
    wits = [Circuit(curve.cid, k, x0=10 + j, x1=j, witness_only=True) for j in range(3)]
    wrong = Circuit(curve.cid, k + 1, witness_only=True)
    keys = Groth16Keys(backend, full, seed=77)
    try:
        seeds = list(range(100, 111))
        circs = [None if i % 4 == 0 else wits[i % 3] for i in range(len(seeds))]
        single = [keys.prove(seed=s, circuit=c)[0] for s, c in zip(seeds, circs)]
        many = keys.prove_many(seeds, [c or full for c in circs])
        assert len(many) == len(single) and all(_same(a, b) for a, b in zip(single, many))
        assert keys.verify(many[1], wits[1].arrays()["assignment"][1:2])
        assert keys.prove_many([]) == [] and _same(keys.prove_many([100])[0], single[0])
        with pytest.raises(BackendError):
            keys.prove_many(seeds[:4], [full, wits[0], wrong, wits[1]])
        assert all(_same(a, b) for a, b in zip(single, keys.prove_many(seeds, [c or full for c in circs])))  # usable after an error
        # the lane the library keeps between prove_many calls is ITS fork, not the caller's: it must not pin the key (a free refused with
        # ZL_EINVAL whose handle the caller then dropped leaked the whole device-resident key).  Freeing the key right after prove_many works and frees it.
        a_q = keys.pk.a_query
        import torch
        free0 = torch.cuda.mem_get_info()[0]
        keys.close()
        with pytest.raises(BackendError):
            backend.bases_download(a_q, 0, 1)     # the handle is gone ...
        assert torch.cuda.mem_get_info()[0] >= free0  # ... and nothing of the key stayed behind
        assert backend.L.zl_ctx_drop_lanes(backend._ctx) == 0
    finally:
        backend.L.zl_ctx_drop_lanes(backend._ctx)
        keys.close()
        for c in [full, wrong] + wits:
            c.close()
