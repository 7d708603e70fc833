"""ProvingContext / VerifyingKey wire formats (/root/reference/plugins/arkworks/src/groth16.rs:142-179): the keys the
backend compiles on the device are encoded exactly as the independent Python restatement (oracle/pyoracle.py) encodes the oracle's own
setup of the same circuit and trapdoor; a context decoded from those bytes proves and verifies like the compiled one.
UNPINNED by arkworks-produced bytes (the reference holds none)."""
import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import Circuit, Groth16Keys
from openzl_amd import backend as zb

pytestmark = pytest.mark.gpu


def _g1_pts(curve, arr):
    arr = np.asarray(arr).reshape(-1, 2 * ol.nlq(curve))
    return [ol.limbs_to_point(curve, row, int(not row.any())) for row in arr]


def _g2_pts(curve, arr):
    nq = ol.nlq(curve)
    out = []
    for row in np.asarray(arr).reshape(-1, 4 * nq):
        if not row.any():
            out.append(None)
            continue
        v = ol.limbs_to_ints(row.reshape(4, nq))
        out.append(((v[0], v[1]), (v[2], v[3])))
    return out


def _oracle_key_points(curve, k, td):
    cs = po.poseidon_chain_circuit(curve.fr, k)
    pk = gu.setup_with_trapdoor(curve, cs, td)
    g1 = lambda ks: _g1_pts(curve, ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ks, 4)))
    vk = {"alpha_g1": _g1_pts(curve, pk["alpha_g1"])[0], "beta_g2": _g2_pts(curve, pk["beta_g2"])[0],
          "gamma_g2": _g2_pts(curve, gu.g2_mul_gen(curve, [td.gamma]))[0], "delta_g2": _g2_pts(curve, pk["delta_g2"])[0],
          "gamma_abc_g1": g1(pk["ex"]["gamma_abc"])}
    return {"vk": vk, "beta_g1": _g1_pts(curve, pk["beta_g1"])[0], "delta_g1": _g1_pts(curve, pk["delta_g1"])[0],
            "a_query": _g1_pts(curve, pk["a_query"]), "b_g1_query": _g1_pts(curve, pk["b_g1_query"]), "b_g2_query": _g2_pts(curve, pk["b_g2_query"]),
            "h_query": _g1_pts(curve, pk["h_query"]), "l_query": _g1_pts(curve, pk["l_query"])}


@pytest.mark.parametrize("curve,k", [(po.BLS12_381, 2), (po.BN254, 1)], ids=["bls-k2", "bn254-k1"])
def test_proving_context_bytes_match_restatement_and_decode_proves(backend, curve, k):
    circ = Circuit(curve.cid, k)
    keys = Groth16Keys(backend, circ, seed=0x5EED06)
    keys2 = None
    try:
        td = po.Groth16Trapdoor(*keys.trapdoor())
        pts = _oracle_key_points(curve, k, td)
        assert any(P is None for P in pts["b_g1_query"]) and any(P is None for P in pts["b_g2_query"])  # variables absent from B: (0, 1) | 0x40 records
        data = keys.to_bytes()
        assert data == po.groth16_pk_bytes(curve, pts)
        assert keys.vk_to_bytes() == po.groth16_vk_bytes(curve, pts["vk"], compressed=True)
        # decode -> a context without trapdoor and without circuit that proves exactly like the compiled one
        keys2 = Groth16Keys.from_bytes(backend, circ, data, check=True)
        with pytest.raises(zb.BackendError):
            keys2.trapdoor()
        proof1, r1, s1 = keys.prove(seed=0xBEEF)
        proof2, r2, s2 = keys2.prove(seed=0xBEEF)
        assert (r1 == r2).all() and (s1 == s2).all()
        assert zb.proof_to_bytes(curve.cid, proof1) == zb.proof_to_bytes(curve.cid, proof2)
        pub = circ.arrays()["assignment"][1:circ.shape[1]]
        assert keys2.verify(proof2, pub) and keys.verify(proof2, pub)
        proof3, _, _ = keys2.prove(seed=0xF00D)  # second proof on the now resident matrices
        assert keys2.verify(proof3, pub)
        assert keys2.to_bytes() == data and keys2.vk_to_bytes() == keys.vk_to_bytes()
        # a circuit of another shape is refused by the decoded key
        other = Circuit(curve.cid, k + 1)
        try:
            keys3 = Groth16Keys.from_bytes(backend, other, data)
            with pytest.raises(zb.BackendError):
                keys3.prove(seed=1)
            keys3.close()
        finally:
            other.close()
    finally:
        if keys2 is not None:
            keys2.close()
        keys.close()
        circ.close()


def test_proving_context_decode_rejects_malformed_input(backend):
    curve = po.BLS12_381
    circ = Circuit(curve.cid, 1)
    keys = Groth16Keys(backend, circ, seed=7)
    try:
        data = keys.to_bytes()
        nb = 48

        def refused(b, check=False, code=-1):
            with pytest.raises(zb.BackendError) as e:
                Groth16Keys.from_bytes(backend, circ, bytes(b), check=check).close()
            assert e.value.code == code

        refused(data[:-1])            # truncated
        refused(data + b"\0")         # trailing byte
        refused(b"")                  # empty
        # the gamma_abc length prefix sits after alpha_g1 (96 B) and three G2 points (3 x 192 B)
        off = 2 * nb + 3 * 4 * nb
        n_abc = int.from_bytes(data[off:off + 8], "little")
        assert n_abc == circ.shape[1]
        bad = bytearray(data); bad[off:off + 8] = (1 << 60).to_bytes(8, "little")
        refused(bad)                  # a length that the input cannot hold
        bad = bytearray(data); bad[off:off + 8] = (n_abc + 1).to_bytes(8, "little")
        refused(bad)                  # shifts every later field: lengths no longer add up
        bad = bytearray(data); bad[0:nb] = curve.fq.p.to_bytes(nb, "little")
        refused(bad)                  # alpha_g1.x = q: not a canonical integer
        # first a_query record: after vk, beta_g1, delta_g1 and the a_query length
        a0 = off + 8 + n_abc * 2 * nb + 2 * 2 * nb + 8
        # take a finite a_query record and move it off the curve: unchecked decode takes it (deserialize_unchecked), ZL_CHECK refuses
        n_a = int.from_bytes(data[a0 - 8:a0], "little")
        i = next(j for j in range(n_a) if data[a0 + j * 2 * nb + 2 * nb - 1] & 0x40 == 0)
        bad = bytearray(data)
        y = int.from_bytes(bad[a0 + i * 2 * nb + nb:a0 + (i + 1) * 2 * nb], "little")
        bad[a0 + i * 2 * nb + nb:a0 + (i + 1) * 2 * nb] = ((y + 1) % curve.fq.p).to_bytes(nb, "little")
        Groth16Keys.from_bytes(backend, circ, bytes(bad), check=False).close()
        refused(bad, check=True, code=-6)
    finally:
        keys.close()
        circ.close()


def test_proving_context_decode_survives_random_corruption(backend):
    """Byte flips anywhere in the encoding: decode either refuses (ZL_EINVAL / ZL_ENOTCURVE) or returns keys that still prove without
    faulting (deserialize_unchecked semantics: canonical garbage coordinates are taken as given) -- never a crash, and with ZL_CHECK a
    flipped coordinate of a finite point is always caught."""
    curve = po.BLS12_381
    circ = Circuit(curve.cid, 1)
    keys = Groth16Keys(backend, circ, seed=11)
    try:
        data = keys.to_bytes()
        rng = np.random.default_rng(20260928)
        accepted = refused = 0
        for trial in range(40):
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            try:
                k2 = Groth16Keys.from_bytes(backend, circ, bytes(bad), check=bool(trial & 1))
            except zb.BackendError as e:
                assert e.code in (-1, -6)
                refused += 1
                continue
            try:
                k2.prove(seed=trial)  # whatever the points are, the prover must come back
                accepted += 1
            finally:
                k2.close()
        assert refused > 0 and accepted + refused == 40
        # the intact encoding still decodes after all of that
        Groth16Keys.from_bytes(backend, circ, data, check=True).close()
    finally:
        keys.close()
        circ.close()
