"""Pairing + Groth16::verify (row f4).  CPU: the C++ host pairing equals the definition-level Python pairing coefficient by
coefficient and is bilinear (mirrors plugins/arkworks/src/pairing.rs:116-129 `*_has_valid_pairing_ratio`: e(g1, s g2) == e(s g1, g2)).
gpu: ProofSystem end to end -- compile, prove on the GPU, verify with the host pairing; tampering is rejected."""
import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import Circuit, Groth16Keys, pairing

CURVES = [po.BLS12_381, po.BN254]


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_host_pairing_matches_definition_and_is_bilinear(curve):
    G1, G2 = po.g1_generator(curve), po.g2_generator(curve)
    s = 0xC0FFEE1234567
    P = ol.points_to_limbs(curve, [G1, po.g1_mul(curve, s, G1)])
    Q = gu.g2_mul_gen(curve, [1, s])
    e_gg = pairing(curve.cid, P[0], Q[0])
    assert ol.limbs_to_ints(e_gg) == po.pairing(curve, G1, G2)
    # same ratio: e(g1, s g2) == e(s g1, g2)
    assert (pairing(curve.cid, P[0], Q[1]) == pairing(curve.cid, P[1], Q[0])).all()
    assert ol.limbs_to_ints(pairing(curve.cid, P[1], Q[0])) == po.Fq12Ctx(curve).pow(po.pairing(curve, G1, G2), s)
    # infinity on either side -> 1
    one = [1] + [0] * 11
    assert ol.limbs_to_ints(pairing(curve.cid, np.zeros_like(P[0]), Q[0])) == one
    assert ol.limbs_to_ints(pairing(curve.cid, P[0], np.zeros_like(Q[0]))) == one


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_pairing_product_in_lock_step_equals_product_of_pairings(curve):
    """Groth16::verify takes its four pairings through ONE lock-step Miller loop (one shared inversion per step, csrc/zl_pairing.h): the product of n pairings
    equals the product of the single pairings (Python Fq12 arithmetic), a pair at infinity drops out, and e(aP, Q) e(P, -aQ) = 1."""
    from openzl_amd.backend import hook_pairing_product

    G1 = po.g1_generator(curve)
    ks, ss = [3, 0xABCDEF12345, 7, 0x1234567], [5, 11, 0xFEDCBA987, 1]
    P = ol.points_to_limbs(curve, [po.g1_mul(curve, k, G1) for k in ks])
    Q = gu.g2_mul_gen(curve, ss)
    ctx = po.Fq12Ctx(curve)
    singles = [ol.limbs_to_ints(pairing(curve.cid, P[i], Q[i])) for i in range(4)]
    exp = singles[0]
    for v in singles[1:]:
        exp = ctx.mul(exp, v)
    assert ol.limbs_to_ints(hook_pairing_product(curve.cid, P, Q)) == exp
    # a pair with the point at infinity contributes 1
    P2 = P.copy()
    P2[2] = 0
    exp2 = ctx.mul(ctx.mul(singles[0], singles[1]), singles[3])
    assert ol.limbs_to_ints(hook_pairing_product(curve.cid, P2, Q)) == exp2
    # e(a P, Q) e(P, -a Q) = 1
    a = 0x5EED5EED5EED
    Pa = ol.points_to_limbs(curve, [po.g1_mul(curve, a, G1), G1])
    Qa = gu.g2_mul_gen(curve, [1, curve.fr.p - a])
    assert ol.limbs_to_ints(hook_pairing_product(curve.cid, Pa, Qa)) == [1] + [0] * 11


@pytest.mark.gpu
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_proof_system_compile_prove_verify(backend, curve):
    circ = Circuit(curve.cid, 2)
    keys = Groth16Keys(backend, circ, seed=0x5EED)
    try:
        proof, r, s = keys.prove(seed=42)
        pub = circ.arrays()["assignment"][1:2]  # the single public input (index 0 is the constant ONE)
        assert keys.verify(proof, pub) is True
        bad_pub = pub.copy()
        bad_pub[0, 0] ^= np.uint64(1)
        assert keys.verify(proof, bad_pub) is False
        a, ai, b, bi, c, ci = proof
        other, _, _ = keys.prove(seed=43)           # different (r, s): a different, equally valid proof
        assert not np.array_equal(other[0], a) and keys.verify(other, pub) is True
        mixed = (a, ai, b, bi, other[4], ci)        # A, B of one proof with C of another
        assert keys.verify(mixed, pub) is False
        # wire format round trip of a device-made proof: 192 / 128 bytes, decodes to the same points, still verifies
        from openzl_amd.backend import proof_from_bytes, proof_to_bytes
        wire = proof_to_bytes(curve.cid, proof)
        assert len(wire) == (192 if curve.cid == 1 else 128)
        back = proof_from_bytes(curve.cid, wire)
        assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(back, proof)) and keys.verify(back, pub) is True
        # cross-check with the definition-level verifier of the oracle (independent pairing implementation)
        td = po.Groth16Trapdoor(*keys.trapdoor())
        cs = po.poseidon_chain_circuit(curve.fr, 2)
        ex = po.groth16_setup_exponents(curve, cs, td)
        G1, G2 = po.g1_generator(curve), po.g2_generator(curve)
        vk = {"alpha_g1": po.g1_mul(curve, td.alpha, G1), "beta_g2": po.g2_mul(curve, td.beta, G2), "gamma_g2": po.g2_mul(curve, td.gamma, G2),
              "delta_g2": po.g2_mul(curve, td.delta, G2), "gamma_abc_g1": [po.g1_mul(curve, e, G1) for e in ex["gamma_abc"]]}
        nq = ol.nlq(curve)
        A = ol.limbs_to_point(curve, a, ai)
        Cp = ol.limbs_to_point(curve, c, ci)
        bl = ol.limbs_to_ints(b.reshape(4, nq))
        B = ((bl[0], bl[1]), (bl[2], bl[3]))
        assert po.groth16_verify_pairing(curve, vk, cs.pub[1:], A, B, Cp)
    finally:
        keys.close()
        circ.close()


@pytest.mark.gpu
def test_large_key_uses_window_tables_and_still_verifies(backend):
    """A key above 2^19 variables: Groth16::compile builds window tables for all five queries (merged bucket sets, sub-range first = 1);
    the proof must verify, a tampered public input must not, and the proof must be reproducible."""
    curve = po.BLS12_381
    circ = Circuit(curve.cid, 2300)  # 538 201 constraints, domain 2^20
    keys = Groth16Keys(backend, circ, seed=0x5EED)
    try:
        assert circ.shape[0] > (1 << 19)
        proof, r, s = keys.prove(seed=5)
        again, _, _ = keys.prove(seed=5)
        assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(proof, again))
        pub = circ.arrays()["assignment"][1:2]
        assert keys.verify(proof, pub) is True
        bad = pub.copy()
        bad[0, 0] ^= np.uint64(1)
        assert keys.verify(proof, bad) is False
    finally:
        keys.close()
        circ.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_pairing_product_falls_back_on_a_vertical_line(curve):
    """the lock-step Miller loops cannot step over a vertical line (one shared inversion of the slope denominators) and hand the whole product to
    the one-by-one loops.  Points of the prime-order subgroups never produce one, so the path is driven with a Q whose y is zero (not a curve point: its first
    tangent is vertical): the product of [a valid pair, the degenerate pair] must still be the Fq12 product of the two single values (the final exponentiation is
    a homomorphism), whichever path computed them -- and a status comes back, not a crash."""
    from openzl_amd.backend import hook_pairing_product

    G1 = po.g1_generator(curve)
    P = ol.points_to_limbs(curve, [po.g1_mul(curve, 9, G1), po.g1_mul(curve, 0x77, G1)])
    Q = gu.g2_mul_gen(curve, [5, 11])
    Qbad = Q.copy()
    half = Qbad.shape[1] // 2
    Qbad[1, half:] = 0  # y = 0 (x kept): tangent at Q is vertical
    ctx = po.Fq12Ctx(curve)
    s0 = ol.limbs_to_ints(pairing(curve.cid, P[0], Q[0]))
    s1 = ol.limbs_to_ints(hook_pairing_product(curve.cid, P[1:2], Qbad[1:2]))
    got = ol.limbs_to_ints(hook_pairing_product(curve.cid, P, Qbad))
    assert got == ctx.mul(s0, s1)
    assert got != ctx.mul(s0, ol.limbs_to_ints(pairing(curve.cid, P[1], Q[1])))  # ... and it is not the value of the valid pair
