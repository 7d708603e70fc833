"""Committed golden vectors (tests/golden/oracle_vectors.json): the CPU test re-derives them with the oracle, the gpu test
runs the HIP path through the C ABI against them."""
import json
import os

import numpy as np
import pytest

import groth16_util as gu
import oracle_lib as ol
from oracle_lib import po

HERE = os.path.dirname(os.path.abspath(__file__))
GV = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))
CURVES = [po.BLS12_381, po.BN254]
TD = lambda v: po.Groth16Trapdoor(*[int(x, 16) for x in v])


def _msm_case(curve):
    g = GV[curve.name]["msm"]
    pts = [None if p is None else (int(p[0], 16), int(p[1], 16)) for p in g["points"]]
    sc = [int(v, 16) for v in g["scalars"]]
    res = (int(g["result"][0], 16), int(g["result"][1], 16))
    return pts, sc, res


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_oracle_reproduces_golden_vectors(curve):
    pts, sc, res = _msm_case(curve)
    assert po.msm_naive(curve, sc, pts) == res
    xy, inf = ol.oracle_msm_g1(curve, ol.points_to_limbs(curve, pts), ol.ints_to_limbs(sc, 4), 0, 1)
    assert ol.limbs_to_point(curve, xy, inf) == res
    g = GV[curve.name]["ntt"]
    x = [int(v, 16) for v in g["input"]]
    for inv in (0, 1):
        for cos in (0, 1):
            exp = [int(v, 16) for v in g[f"inverse{inv}_coset{cos}"]]
            assert ol.limbs_to_ints(ol.oracle_ntt(curve, ol.ints_to_limbs(x, 4), bool(inv), bool(cos))) == exp


@pytest.mark.gpu
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_gpu_matches_golden_vectors(backend, curve):
    pts, sc, res = _msm_case(curve)
    h = backend.bases_upload(curve.cid, ol.points_to_limbs(curve, pts))
    got, inf = backend.msm(h, ol.ints_to_limbs(sc, 4))
    backend.bases_free(h)
    assert ol.limbs_to_point(curve, got, inf) == res
    g = GV[curve.name]["ntt"]
    x = ol.ints_to_limbs([int(v, 16) for v in g["input"]], 4)
    for inv in (0, 1):
        for cos in (0, 1):
            exp = [int(v, 16) for v in g[f"inverse{inv}_coset{cos}"]]
            assert ol.limbs_to_ints(backend.ntt(curve.cid, x, inverse=bool(inv), coset=bool(cos))) == exp
    # Groth16: one Poseidon hash with the fixture's trapdoor, r, s
    gg = GV[curve.name]["groth16_poseidon_k1"]
    cs = po.poseidon_chain_circuit(curve.fr, 1)
    assert cs.n_constraints == gg["constraints"]
    pk = gu.setup_with_trapdoor(curve, cs, TD(gg["trapdoor"]))
    dpk = gu.upload_pk(backend, curve, pk)
    try:
        a, ai, b, bi, c, ci = backend.groth16_prove(curve.cid, dpk, gu.r1cs_arrays(cs), ol.ints_to_limbs(cs.assignment(), 4),
                                                    ol.ints_to_limbs([int(gg["r"], 16)], 4)[0], ol.ints_to_limbs([int(gg["s"], 16)], 4)[0])
    finally:
        gu.free_pk(backend, dpk)
    nq = ol.nlq(curve)
    assert not (ai or bi or ci)
    assert [hex(v) for v in ol.limbs_to_ints(a.reshape(2, nq))] == gg["proof_a"]
    assert [hex(v) for v in ol.limbs_to_ints(b.reshape(4, nq))] == gg["proof_b"]
    assert [hex(v) for v in ol.limbs_to_ints(c.reshape(2, nq))] == gg["proof_c"]
