"""The C++ host mirror of the plugin interface (openzl::poseidon / R1CS<F> / Groth16<E>, csrc/zl_host.h).
CPU tests: the product's native Poseidon reproduces the reference's own KAT, and its R1CS compiler builds exactly the
circuit the independent Python builder (oracle) builds.  gpu test: Groth16::compile + prove end to end."""
import json
import os

import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import Circuit, Groth16Keys, poseidon_permute

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "ref_poseidon_fixtures.json")))


def test_native_poseidon_matches_reference_kat():
    """mirrors openzl-tutorials/src/poseidon.rs:364-405 `poseidon_arity_2` (same input, same three expected values)"""
    out = poseidon_permute(po.BLS12_381.cid, ol.ints_to_limbs([3, 1, 2], 4))
    assert [str(v) for v in ol.limbs_to_ints(out)] == FX["permutation_width3"]["output"]


def test_native_poseidon_bn254_matches_python_model():
    out = poseidon_permute(po.BN254.cid, ol.ints_to_limbs([3, 1, 2], 4))
    assert ol.limbs_to_ints(out) == po.poseidon_permute(po.BN254_FR, [3, 1, 2])


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("k", [1, 2, 3, 6])
def test_r1cs_compiler_builds_the_same_circuit_as_the_python_builder(curve, k):
    c = Circuit(curve.cid, k)
    cs = po.poseidon_chain_circuit(curve.fr, k)
    assert c.shape == (cs.n_constraints, cs.n_instance, cs.n_witness) == (234 * k + 1, 2, 234 * k + 2)
    assert c.is_satisfied() and cs.is_satisfied()
    got, exp = c.arrays(), gu.r1cs_arrays(cs)
    for key in "ABC":
        for g, e in zip(got[key], exp[key]):
            assert np.array_equal(g, e)
    assert np.array_equal(got["assignment"], ol.ints_to_limbs(cs.assignment(), 4))
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve,k", [(po.BLS12_381, 2), (po.BN254, 1)], ids=["bls-k2", "bn254-k1"])
def test_groth16_compile_and_prove_end_to_end(backend, curve, k):
    circ = Circuit(curve.cid, k)
    keys = Groth16Keys(backend, circ, seed=0xC0FFEE)
    try:
        (a, ai, b, bi, c, ci), r, s = keys.prove(seed=0xBEEF)
        arrays = circ.arrays()
        n = 1 << max(1, (arrays["n_constraints"] + arrays["n_instance"] - 1).bit_length())
        h_gpu = backend.groth16_last_h(n)
        # (1) the setup: proving key = exponents * generator, recomputed independently from the trapdoor
        td = po.Groth16Trapdoor(*keys.trapdoor())
        cs = po.poseidon_chain_circuit(curve.fr, k)
        ex = po.groth16_setup_exponents(curve, cs, td)
        pkd = keys.pk_dict()
        assert (backend.bases_download(pkd["a_query"]) == ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ex["a_query"], 4))).all()
        assert (backend.bases_download(pkd["l_query"]) == ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ex["l_query"], 4))).all()
        assert (backend.bases_download(pkd["h_query"]) == ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ex["h_query"], 4))).all()
        assert (backend.bases_download(pkd["b_g2_query"]) == gu.g2_mul_gen(curve, ex["b_query"])).all()
        # (2) the proof: recomputed in the exponent and checked against the Groth16 verification equation
        h_py = po.qap_witness_map(curve, cs)
        assert ol.limbs_to_ints(h_gpu) == h_py
        A, B, Cx = po.groth16_prove_exponents(curve, cs, td, ex, h_py, ol.limbs_to_ints(r.reshape(1, 4))[0], ol.limbs_to_ints(s.reshape(1, 4))[0])
        assert po.groth16_check_exponents(curve, cs, td, ex, A, B, Cx)
        assert not (ai or bi or ci)
        assert (a == ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([A], 4))[0]).all()
        assert (b == gu.g2_mul_gen(curve, [B])[0]).all()
        assert (c == ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs([Cx], 4))[0]).all()
    finally:
        keys.close()
        circ.close()


@pytest.mark.gpu
def test_proving_context_refuses_a_different_circuit_of_the_same_shape(backend):
    """The constraint matrices are device-resident per proving context and only the assignment travels per proof, so the context is bound to
    its circuit by a fingerprint of the matrices (R1CS::structure_digest): the same circuit with another witness proves, a circuit of the same
    shape with one coefficient changed (test hook zl_test_circuit_tweak) is refused -- for compiled keys and for keys decoded from bytes, which
    bind at their first proof (a decoded key silently kept the first circuit's matrices)."""
    from openzl_amd import BackendError, Circuit, Groth16Keys

    curve = po.BLS12_381
    c1 = Circuit(curve.cid, 2)
    keys = Groth16Keys(backend, c1, seed=11)
    c2 = Circuit(curve.cid, 2, x0=5, x1=9)      # same circuit, other witness
    c3 = Circuit(curve.cid, 2)
    assert c3.L.zl_test_circuit_tweak(c3._c) == 0  # same shape, different matrices
    try:
        p1, _, _ = keys.prove(seed=3)
        assert keys.verify(p1, c1.arrays()["assignment"][1:2])
        old = keys.circuit
        keys.circuit = c2
        p2, _, _ = keys.prove(seed=3)
        assert keys.verify(p2, c2.arrays()["assignment"][1:2])
        keys.circuit = c3
        with pytest.raises(BackendError) as e:
            keys.prove(seed=3)
        assert e.value.code == -1
        keys.circuit = old
        # a decoded key binds to the first circuit it proves
        dec = Groth16Keys.from_bytes(backend, c1, keys.to_bytes())
        try:
            dec.prove(seed=4)
            dec.circuit = c3
            with pytest.raises(BackendError):
                dec.prove(seed=4)
        finally:
            dec.close()
    finally:
        keys.close()
        for c in (c1, c2, c3):
            c.close()


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_witness_only_compiler_yields_the_same_assignment(curve):
    """R1CS::for_witness (ark-relations' SynthesisMode::Prove { construct_matrices: false }): the circuit code runs, every variable gets its value, no linear
    combination is formed and no row stored -- the assignment must be the full compiler's, element for element, and the enforced equality must hold."""
    for k, x0, x1 in ((1, 1, 2), (3, 7, 0), (5, curve.fr.p - 1, 12345)):
        full, wit = Circuit(curve.cid, k, x0=x0, x1=x1), Circuit(curve.cid, k, x0=x0, x1=x1, witness_only=True)
        try:
            fa, wa = full.arrays(), wit.arrays()
            assert wit.shape == (0, full.shape[1], full.shape[2]) and wit.is_satisfied() and full.is_satisfied()
            assert np.array_equal(fa["assignment"], wa["assignment"])
            assert all(wa[m][0].tolist() == [0] and wa[m][1].size == 0 for m in "ABC")
        finally:
            full.close()
            wit.close()


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_prove_with_a_witness_only_compiler(backend, curve):
    """A context that holds its circuit's matrices proves from a witness-only compiler: the same proof as from the full compiler for the same witness and rng,
    a verifying proof for a NEW witness (what a prover does per proof: the matrices are static), and refusals where the binding cannot hold: another shape,
    or a context decoded from bytes that has not seen a full compiler yet."""
    from openzl_amd import BackendError

    k = 3
    full = Circuit(curve.cid, k)
    keys = Groth16Keys(backend, full, seed=21)
    w_same = Circuit(curve.cid, k, witness_only=True)
    w_new = Circuit(curve.cid, k, x0=99, x1=77, witness_only=True)
    w_shape = Circuit(curve.cid, k + 1, witness_only=True)
    decoded = None
    try:
        p_full, r1, s1 = keys.prove(seed=5)
        p_wit, r2, s2 = keys.prove(seed=5, circuit=w_same)
        assert (r1 == r2).all() and (s1 == s2).all()
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p_full, p_wit))
        p_new, _, _ = keys.prove(seed=6, circuit=w_new)
        pub_new = w_new.arrays()["assignment"][1:2]
        assert keys.verify(p_new, pub_new) and not keys.verify(p_new, full.arrays()["assignment"][1:2])
        with pytest.raises(BackendError):
            keys.prove(seed=7, circuit=w_shape)
        decoded = Groth16Keys.from_bytes(backend, full, keys.to_bytes())
        with pytest.raises(BackendError):
            decoded.prove(seed=8, circuit=w_same)       # unbound: no rows to bind to
        decoded.prove(seed=8)                            # the full compiler binds it ...
        p_dec, _, _ = decoded.prove(seed=5, circuit=w_same)  # ... and then the witness-only one proves
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p_full, p_dec))
    finally:
        if decoded is not None:
            decoded.close()
        keys.close()
        for c in (full, w_same, w_new, w_shape):
            c.close()
