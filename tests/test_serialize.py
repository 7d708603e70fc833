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
