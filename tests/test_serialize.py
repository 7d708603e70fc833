"""Wire format: csrc/zl_serialize.h (arkworks 0.3 CanonicalSerialize, compressed) against the independent Python
restatement in oracle/pyoracle.py, plus round trips and malformed inputs.  Host code only (no GPU).  The format itself is
UNPINNED by reference vectors (the reference holds none); these tests pin the C++ to the restatement."""
import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import backend as zb

CURVES = [po.BLS12_381, po.BN254]


def _g1_points(curve, n, seed):
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, seed))
    return [po.g1_mul(curve, k, po.g1_generator(curve)) for k in ks]


def _g2_points(curve, n, seed):
    ks = ol.limbs_to_ints(ol.random_scalars(curve, n, seed))
    return [po.g2_mul(curve, k, po.g2_generator(curve)) for k in ks]


def _g1_limbs(curve, P):
    nq = ol.nlq(curve)
    return np.array(ol.ints_to_limbs([P[0], P[1]], nq)).reshape(-1) if P is not None else np.zeros(2 * nq, dtype=np.uint64)


def _g2_limbs(curve, P):
    nq = ol.nlq(curve)
    if P is None:
        return np.zeros(4 * nq, dtype=np.uint64)
    return np.array(ol.ints_to_limbs([P[0][0], P[0][1], P[1][0], P[1][1]], nq)).reshape(-1)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_sizes(curve):
    L = zb.load_library()
    nb = 48 if curve.cid == 1 else 32
    assert L.zl_point_bytes(curve.cid, 1) == nb and L.zl_point_bytes(curve.cid, 2) == 2 * nb
    assert L.zl_groth16_proof_bytes(curve.cid) == 4 * nb  # 192 / 128


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_g1_matches_restatement_and_round_trips(curve):
    for P in _g1_points(curve, 6, 11) + [None, po.g1_generator(curve), po.g1_neg(curve, po.g1_generator(curve))]:
        data = zb.point_to_bytes(curve.cid, 1, _g1_limbs(curve, P), inf=int(P is None))
        assert data == po.g1_compress(curve, P)
        assert po.g1_decompress(curve, data) == P
        xy, inf = zb.point_from_bytes(curve.cid, 1, data)
        assert inf == int(P is None) and (xy == _g1_limbs(curve, P)).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_g2_matches_restatement_and_round_trips(curve):
    G = po.g2_generator(curve)
    negG = (G[0], ((curve.fq.p - G[1][0]) % curve.fq.p, (curve.fq.p - G[1][1]) % curve.fq.p))
    for P in _g2_points(curve, 4, 12) + [None, G, negG]:
        data = zb.point_to_bytes(curve.cid, 2, _g2_limbs(curve, P), inf=int(P is None))
        assert data == po.g2_compress(curve, P)
        assert po.g2_decompress(curve, data) == P
        xy, inf = zb.point_from_bytes(curve.cid, 2, data)
        assert inf == int(P is None) and (xy == _g2_limbs(curve, P)).all()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_proof_bytes(curve):
    A, C_ = _g1_points(curve, 2, 21)
    B = _g2_points(curve, 1, 22)[0]
    proof = (_g1_limbs(curve, A), 0, _g2_limbs(curve, B), 0, _g1_limbs(curve, C_), 0)
    data = zb.proof_to_bytes(curve.cid, proof)
    assert data == po.groth16_proof_bytes(curve, A, B, C_) and len(data) == (192 if curve.cid == 1 else 128)
    back = zb.proof_from_bytes(curve.cid, data)
    for got, exp in zip(back, proof):
        assert np.array_equal(np.asarray(got), np.asarray(exp))


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_malformed_inputs(curve):
    nb = 48 if curve.cid == 1 else 32
    good = zb.point_to_bytes(curve.cid, 1, _g1_limbs(curve, po.g1_generator(curve)))
    with pytest.raises(zb.BackendError):  # wrong length
        zb.proof_from_bytes(curve.cid, good)
    both = bytearray(good)
    both[-1] |= 0xC0  # both flags
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes(curve.cid, 1, bytes(both))
    inf_nonzero = bytearray(good)
    inf_nonzero[-1] = (inf_nonzero[-1] & 0x3F) | 0x40  # infinity flag with x != 0
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes(curve.cid, 1, bytes(inf_nonzero))
    q_bytes = bytearray(curve.fq.p.to_bytes(nb, "little"))  # x = q is not canonical
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes(curve.cid, 1, bytes(q_bytes))
    # an x with no point on the curve: scan small x until x^3 + b is a non-residue
    x = 1
    while po.fq_sqrt(curve.fq.p, (x ** 3 + curve.b) % curve.fq.p) is not None:
        x += 1
    with pytest.raises(zb.BackendError) as e:
        zb.point_from_bytes(curve.cid, 1, x.to_bytes(nb, "little"))
    assert e.value.code == -6  # ZL_ENOTCURVE


def test_bls_g1_point_outside_the_subgroup_is_rejected():
    """BLS12-381 G1 has a cofactor: a curve point of non-prime order must not deserialize (ark-ec checks the subgroup too)."""
    curve = po.BLS12_381
    p = curve.fq.p
    x = 1
    while True:
        y = po.fq_sqrt(p, (x ** 3 + curve.b) % p)
        # r * P computed as (r - 1) P + P: the oracle's g1_mul reduces its scalar modulo r
        if y is not None and po.g1_add(curve, po.g1_mul(curve, curve.fr.p - 1, (x, y)), (x, y)) is not None:
            break
        x += 1
    data = bytearray(x.to_bytes(48, "little"))
    if y > p - y:
        data[-1] |= 0x80
    with pytest.raises(zb.BackendError) as e:
        zb.point_from_bytes(curve.cid, 1, bytes(data))
    assert e.value.code == -6


# ---- uncompressed form (serialize_uncompressed / serialize_unchecked: what the ProvingContext codec writes, groth16.rs:142-179) ----------
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_uncompressed_points_match_restatement_and_round_trip(curve):
    nb = 48 if curve.cid == 1 else 32
    L = zb.load_library()
    assert L.zl_point_bytes_uncompressed(curve.cid, 1) == 2 * nb and L.zl_point_bytes_uncompressed(curve.cid, 2) == 4 * nb
    for P in _g1_points(curve, 4, 31) + [None, po.g1_generator(curve)]:
        data = zb.point_to_bytes_uncompressed(curve.cid, 1, _g1_limbs(curve, P), inf=int(P is None))
        assert data == po.g1_uncompressed(curve, P)
        for check in (False, True):
            xy, inf = zb.point_from_bytes_uncompressed(curve.cid, 1, data, check=check)
            assert inf == int(P is None) and (xy == _g1_limbs(curve, P)).all()
    for P in _g2_points(curve, 3, 32) + [None, po.g2_generator(curve)]:
        data = zb.point_to_bytes_uncompressed(curve.cid, 2, _g2_limbs(curve, P), inf=int(P is None))
        assert data == po.g2_uncompressed(curve, P)
        for check in (False, True):
            xy, inf = zb.point_from_bytes_uncompressed(curve.cid, 2, data, check=check)
            assert inf == int(P is None) and (xy == _g2_limbs(curve, P)).all()
    # a finite point carries no flag bits; the infinity record is (0, 1) with bit 6 of the last byte
    assert po.g1_uncompressed(curve, po.g1_generator(curve))[-1] & 0xC0 == 0
    z = po.g1_uncompressed(curve, None)
    assert z[:nb] == bytes(nb) and z[nb] == 1 and z[-1] == 0x40


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_uncompressed_malformed_and_checked(curve):
    nb, p = (48 if curve.cid == 1 else 32), curve.fq.p
    G = po.g1_generator(curve)
    good = bytearray(po.g1_uncompressed(curve, G))
    bad = bytearray(good); bad[-1] |= 0xC0  # both flags
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes_uncompressed(curve.cid, 1, bytes(bad))
    bad = bytearray(good); bad[nb - 1] |= 0x80  # flag bits on x
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes_uncompressed(curve.cid, 1, bytes(bad))
    bad = bytearray(p.to_bytes(nb, "little") + good[nb:])  # x = q is not canonical
    with pytest.raises(zb.BackendError):
        zb.point_from_bytes_uncompressed(curve.cid, 1, bytes(bad))
    with pytest.raises(zb.BackendError):  # wrong length
        zb.point_from_bytes_uncompressed(curve.cid, 1, bytes(good[:-1]))
    # unchecked takes any canonical pair (deserialize_unchecked), checked rejects a pair that is not on the curve
    off = G[0].to_bytes(nb, "little") + ((G[1] + 1) % p).to_bytes(nb, "little")
    xy, inf = zb.point_from_bytes_uncompressed(curve.cid, 1, off, check=False)
    assert inf == 0
    with pytest.raises(zb.BackendError) as e:
        zb.point_from_bytes_uncompressed(curve.cid, 1, off, check=True)
    assert e.value.code == -6
    # an infinity record ignores its coordinates
    junk = bytearray(good); junk[-1] |= 0x40
    xy, inf = zb.point_from_bytes_uncompressed(curve.cid, 1, bytes(junk))
    assert inf == 1 and not xy.any()


# ---- ProvingContext bytes on the host only (zl_groth16_keys_parse): the decoder of untrusted input, without a device -------------------------------
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


def _oracle_key_bytes(curve, k=1):
    """the oracle's own setup of the k-hash Poseidon chain under a fixed trapdoor, encoded by the independent Python restatement (oracle/pyoracle.py) -- no GPU"""
    td = po.Groth16Trapdoor(alpha=11, beta=13, gamma=17, delta=19, tau=23)
    cs = po.poseidon_chain_circuit(curve.fr, k)
    pk = gu.setup_with_trapdoor(curve, cs, td)
    g1 = lambda ks: _g1_pts(curve, ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ks, 4)))
    vk = {"alpha_g1": _g1_pts(curve, pk["alpha_g1"])[0], "beta_g2": _g2_pts(curve, pk["beta_g2"])[0],
          "gamma_g2": _g2_pts(curve, gu.g2_mul_gen(curve, [td.gamma]))[0], "delta_g2": _g2_pts(curve, pk["delta_g2"])[0],
          "gamma_abc_g1": g1(pk["ex"]["gamma_abc"])}
    pts = {"vk": vk, "beta_g1": _g1_pts(curve, pk["beta_g1"])[0], "delta_g1": _g1_pts(curve, pk["delta_g1"])[0],
           "a_query": _g1_pts(curve, pk["a_query"]), "b_g1_query": _g1_pts(curve, pk["b_g1_query"]), "b_g2_query": _g2_pts(curve, pk["b_g2_query"]),
           "h_query": _g1_pts(curve, pk["h_query"]), "l_query": _g1_pts(curve, pk["l_query"])}
    return po.groth16_pk_bytes(curve, pts)


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_proving_context_parse_on_the_host_accepts_the_restatement_and_survives_corruption(curve):
    """zl_groth16_keys_parse reads every byte zl_groth16_keys_from_bytes reads, on the host alone: the Python restatement's encoding of the oracle's key is
    accepted (with and without ZL_CHECK); truncations, trailing bytes, lying Vec lengths and non-canonical coordinates are refused with ZL_EINVAL; 300 random byte
    flips / truncations / splices never crash (this test also runs under AddressSanitizer + UBSan: tests/test_sanitizers.py)."""
    data = _oracle_key_bytes(curve)
    assert zb.groth16_keys_parse(curve.cid, data) == 0
    assert zb.groth16_keys_parse(curve.cid, data, check=True) == 0
    assert zb.groth16_keys_parse(curve.cid, data[:-1]) == -1
    assert zb.groth16_keys_parse(curve.cid, data + b"\0") == -1
    assert zb.groth16_keys_parse(curve.cid, b"") == -1
    assert zb.groth16_keys_parse(3, data) == -1
    nb = (curve.fq.p.bit_length() + 63) // 64 * 8
    off = 2 * nb + 3 * 4 * nb  # gamma_abc length prefix: after alpha_g1 and three G2 points
    for lie in (1 << 60, (1 << 64) - 1, len(data), int.from_bytes(data[off:off + 8], "little") + 1):
        bad = bytearray(data)
        bad[off:off + 8] = lie.to_bytes(8, "little")
        assert zb.groth16_keys_parse(curve.cid, bytes(bad)) == -1, lie
    bad = bytearray(data)
    bad[0:nb] = curve.fq.p.to_bytes(nb, "little")  # alpha_g1.x = q: not canonical
    assert zb.groth16_keys_parse(curve.cid, bytes(bad)) == -1
    bad = bytearray(data)
    y = int.from_bytes(bad[nb:2 * nb], "little")
    bad[nb:2 * nb] = ((y + 1) % curve.fq.p).to_bytes(nb, "little")  # alpha_g1 off the curve: taken as given unchecked, refused under ZL_CHECK
    assert zb.groth16_keys_parse(curve.cid, bytes(bad)) == 0 and zb.groth16_keys_parse(curve.cid, bytes(bad), check=True) == -6
    rng = np.random.default_rng(20260929 + curve.cid)
    seen = set()
    for trial in range(300):
        bad = bytearray(data)
        kind = trial % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(0, len(bad)))] ^= int(rng.integers(1, 256))
        elif kind == 1:
            bad = bad[:int(rng.integers(0, len(bad)))]
        elif kind == 2:
            i, j = sorted(int(v) for v in rng.integers(0, len(bad), 2))
            bad = bad[:i] + bad[j:]
        else:
            i = int(rng.integers(0, len(bad) - 8))
            bad[i:i + 8] = rng.bytes(8)
        seen.add(zb.groth16_keys_parse(curve.cid, bytes(bad), check=bool(trial & 4)))
    assert seen <= {0, -1, -6} and -1 in seen
