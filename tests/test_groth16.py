"""Groth16 prove (config 5: Poseidon-hash chain circuit): CPU tests pin the oracle against the Groth16 equation in
the exponent (known trapdoor); the gpu test checks zl_groth16_prove bit-exact vs the oracle."""
import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po

TD = po.Groth16Trapdoor(alpha=0x1111_2222_3333, beta=0x4444_5555_6666, gamma=0x7777_8888, delta=0x9999_AAAA_BBBB, tau=0xCCCC_DDDD_EEEE_F001)
R_, S_ = 0x1234_5678_9ABC_DEF0_1111, 0x0FED_CBA9_8765_4321_2222


def _case(curve, k):
    cs = po.poseidon_chain_circuit(curve.fr, k)
    assert cs.is_satisfied()
    pk = gu.setup_with_trapdoor(curve, cs, TD)
    arrays = gu.r1cs_arrays(cs)
    z = ol.ints_to_limbs(cs.assignment(), 4)
    r = ol.ints_to_limbs([R_], 4)[0]
    s = ol.ints_to_limbs([S_], 4)[0]
    return cs, pk, arrays, z, r, s


def _expected_points(curve, cs, pk, h_ints):
    A, B, Cx = po.groth16_prove_exponents(curve, cs, TD, pk["ex"], h_ints, R_, S_)
    assert po.groth16_check_exponents(curve, cs, TD, pk["ex"], A, B, Cx)  # e(A,B) = e(alpha,beta) e(pub,gamma) e(C,delta)
    ga = ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([A], 4))[0]
    gc = ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([Cx], 4))[0]
    gb = gu.g2_mul_gen(curve, [B])[0]
    return ga, gb, gc


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_circuit_shape_and_oracle_proof_satisfies_groth16_equation(curve):
    cs, pk, arrays, z, r, s = _case(curve, 1)
    assert (cs.n_constraints, cs.n_instance, cs.n_witness) == (235, 2, 236)  # 79 S-boxes x 3 - 3 folded constants + 1 output
    n = 1 << cs.domain_log()
    (a, ai, b, bi, c, ci), h = gu.oracle_prove(curve, arrays, z, pk, r, s, threads=4, want_h=n)
    h_py = po.qap_witness_map(curve, cs)
    assert ol.limbs_to_ints(h) == h_py and h_py[-1] == 0
    ga, gb, gc = _expected_points(curve, cs, pk, h_py)
    assert not (ai or bi or ci)
    assert (a == ga).all() and (b == gb).all() and (c == gc).all()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,k", [(po.BLS12_381, 1), (po.BN254, 1), (po.BLS12_381, 8), (po.BN254, 8), (po.BLS12_381, 64)],
                         ids=["bls-k1", "bn254-k1", "bls-k8", "bn254-k8", "bls-k64"])  # k = 64: BASELINE config 5's middle size (N = 2^14)
def test_gpu_groth16_prove_matches_oracle(backend, curve, k):
    cs, pk, arrays, z, r, s = _case(curve, k)
    n = 1 << cs.domain_log()
    dpk = gu.upload_pk(backend, curve, pk)
    try:
        got = backend.groth16_prove(curve.cid, dpk, arrays, z, r, s)
        h_gpu = backend.groth16_last_h(n)
    finally:
        gu.free_pk(backend, dpk)
    exp, h = gu.oracle_prove(curve, arrays, z, pk, r, s, threads=8, want_h=n)
    assert (h_gpu == h).all()
    for g, e in zip(got, exp):
        assert np.array_equal(np.asarray(g), np.asarray(e))
    if k == 1:
        ga, gb, gc = _expected_points(curve, cs, pk, ol.limbs_to_ints(h))
        assert (got[0] == ga).all() and (got[2] == gb).all() and (got[4] == gc).all()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_gpu_groth16_folded_c_query_gives_the_same_proof(backend, curve, monkeypatch):
    """Small proofs take C from ONE MSM over l | a | b1 | h with the scalars z_w | s z | r z | h (zl_groth16.hip, G16KeyCache) and the key's
    fixed-base tables; the four-MSM form with s A + r B1 on the host stays for large ones.  Same group elements either way: the proof bytes
    of both forms are equal to each other and to the oracle's, for ordinary and for degenerate blinding scalars, alternating over one key."""
    cs, pk, arrays, z, r, s = _case(curve, 8)
    zero = ol.ints_to_limbs([0], 4)[0]
    rmax = ol.ints_to_limbs([curve.fr.p - 1], 4)[0]
    dpk = gu.upload_pk(backend, curve, pk)
    try:
        for rr, ss in ((r, s), (zero, zero), (rmax, s), (r, zero)):
            exp, _ = gu.oracle_prove(curve, arrays, z, pk, rr, ss, threads=8, want_h=1 << cs.domain_log())
            for mode in ("0", "20", "0", "20"):
                monkeypatch.setenv("ZL_TUNE_G16_FOLD_LOG_N", mode)
                got = backend.groth16_prove(curve.cid, dpk, arrays, z, rr, ss)
                for g, e in zip(got, exp):
                    assert np.array_equal(np.asarray(g), np.asarray(e)), mode
    finally:
        gu.free_pk(backend, dpk)
