"""CPU tests of the oracle itself: pinned to every known-answer vector the reference holds for this path
(BLS12-381 Fr Poseidon fixtures) and to the constants of SURVEY.md §8c.2-3 / Appendix A."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po

HERE = os.path.dirname(os.path.abspath(__file__))
FX = json.load(open(os.path.join(HERE, "golden", "ref_poseidon_fixtures.json")))
F = po.BLS12_381_FR


# ---- reference fixtures (plugins/arkworks/src/poseidon/*_hardcoded_test*, openzl-tutorials poseidon_arity_2) --------
@pytest.mark.parametrize("t", range(2, 13))
def test_py_mds_matches_reference_fixture(t):
    assert [[str(v) for v in row] for row in po.poseidon_mds(F, t)] == FX["mds"][str(t)]


def test_py_lfsr_round_constants_match_reference_fixture():
    assert [str(v) for v in po.poseidon_round_constants(F, 3, 8, 55)] == FX["lfsr_values"]


def test_py_permutation_matches_reference_kat():
    assert [str(v) for v in po.poseidon_permute(F, [3, 1, 2])] == FX["permutation_width3"]["output"]


def test_c_permutation_matches_reference_kat():
    """pins the C oracle's Fr Montgomery mul/add over 63 rounds (thousands of multiplications)"""
    keys = ol.ints_to_limbs([int(v) for v in FX["lfsr_values"]], 4)
    mds = ol.ints_to_limbs([int(v) for row in FX["mds"]["3"] for v in row], 4)
    st = ol.ints_to_limbs([3, 1, 2], 4)
    assert ol.lib().zlo_poseidon3(ol.p64(keys), ol.p64(mds), 8, 55, ol.p64(st)) == 0
    assert [str(v) for v in ol.limbs_to_ints(st)] == FX["permutation_width3"]["output"]


@pytest.mark.parametrize("t", [2, 3, 7, 12])
def test_c_field_inverse_matches_reference_mds(t):
    for i in range(t):
        for j in range(t):
            a = ol.ints_to_limbs([i + t + j], 4)
            r = np.zeros((1, 4), dtype=np.uint64)
            assert ol.lib().zlo_field_op(2, 3, ol.p64(a), ol.p64(a), ol.p64(r)) == 0
            assert str(ol.limbs_to_ints(r)[0]) == FX["mds"][str(t)][i][j]


# ---- SURVEY.md §8c.2-3 / Appendix A constants -----------------------------------------------------------------------
def test_curve_constants():
    c = po.BLS12_381
    G = po.g1_generator(c)
    assert po.g1_is_on_curve(c, G) and po.g1_mul(c, c.fr.p, G) is None
    assert po.g1_add(c, G, G)[0] == 0x0572CBEA904D67468808C8EB50A9450C9721DB309128012543902D0AC358A62AE28F75BB8F1C7C42C39A8C5529BF0F4E
    pts = [po.g1_mul(c, k, G) for k in (1, 2, 3, 4)]
    r = po.msm_naive(c, [1, 2, 3, 4], pts)
    assert r[0] == 0x0D84464B3966EC5BEDE84AA487FACFCA7823AF383715078DA03B387CC2F5D5597CDD7D025AA07DB00A38B953BDEB6E3F
    assert r[1] == 0x174A09CD44CCB04C382893C3D197578C85F48CC7C2AE8BBFC7A8CDA72EE7CA7833DE357666E979A5A9EBAC59B5E1D15D
    b = po.BN254
    assert po.g1_mul(b, 30, po.g1_generator(b)) == (
        1527465159374431915328497116935179161014331322368960485951268517950184093102,
        17274044707157828649723710289902216429715848248207037129568326237800068062774)
    assert po.g1_mul(b, b.fr.p, po.g1_generator(b)) is None
    assert c.two_adic_root == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert b.two_adic_root == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert (c.fq.inv64, c.fr.inv64, b.fq.inv64, b.fr.inv64) == (0x89F3FFFCFFFCFFFD, 0xFFFFFFFEFFFFFFFF, 0x87D20782E4866389, 0xC2E1F593EFFFFFFF)
    for cc in (c, b):
        assert po.g2_is_on_curve(cc, po.g2_generator(cc)) and po.g2_mul(cc, cc.fr.p, po.g2_generator(cc)) is None


def test_ntt_small_known_answers():
    c = po.BLS12_381
    assert po.domain_root(c, 2) == 3465144826073652318776269530687742778270252468765361963008
    assert po.ntt(c, [1, 2, 3, 4]) == [
        10, 52435875175126190472517450856038661200138013439152152266063153762407857258495,
        52435875175126190479447740508185965837690552500527637822603658699938581184511,
        6930289652147304637552539061375485556540504937530723926014]
    assert po.ntt(c, [1, 2, 3, 4], coset=True) == [
        1534, 52435875175126185773781066700166116939516529826572944931600806116577035419503,
        52435875175126190479447740508185965837690552500527637822603658699938581183275,
        4705666673808019848898174022673954692891002852583361545764718]


# ---- C oracle vs the definition-level Python model ------------------------------------------------------------------
@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_c_msm_matches_python_definition(curve):
    rng = po.SplitMix64(1234)
    ks = [po.sample_fr(curve.fr, rng) for _ in range(40)]
    G = po.g1_generator(curve)
    pts = [po.g1_mul(curve, k, G) for k in ks]
    assert (ol.oracle_g1_mul_gen(curve, ol.ints_to_limbs(ks, 4)) == ol.points_to_limbs(curve, pts)).all()
    sc = [po.sample_fr(curve.fr, rng) for _ in range(40)]
    sc[3], sc[4], sc[5], sc[6] = 0, 1, curve.fr.p - 1, 1
    pts[7] = None
    pts[8] = pts[9]
    pts[10] = po.g1_neg(curve, pts[11])
    sc[10] = sc[11]
    exp = po.msm_naive(curve, sc, pts)
    assert po.msm_pippenger_ark(curve, sc, pts) == exp
    B, S = ol.points_to_limbs(curve, pts), ol.ints_to_limbs(sc, 4)
    for algo in (0, 1):
        for th in (1, 4):
            xy, inf = ol.oracle_msm_g1(curve, B, S, algo, th)
            assert ol.limbs_to_point(curve, xy, inf) == exp
    # n < 32 (c = 3 branch) and n = 0
    xy, inf = ol.oracle_msm_g1(curve, B[:5], S[:5], 0, 1)
    assert ol.limbs_to_point(curve, xy, inf) == po.msm_naive(curve, sc[:5], pts[:5])
    xy, inf = ol.oracle_msm_g1(curve, B[:0], S[:0], 0, 1)
    assert inf == 1


@pytest.mark.parametrize("curve", [po.BLS12_381, po.BN254], ids=lambda c: c.name)
def test_c_ntt_matches_python_definition(curve):
    rng = po.SplitMix64(99)
    x = [po.sample_fr(curve.fr, rng) for _ in range(64)]
    for inv in (False, True):
        for cos in (False, True):
            e = po.dft_naive(curve, x, inv, cos)
            g = ol.limbs_to_ints(ol.oracle_ntt(curve, ol.ints_to_limbs(x, 4), inv, cos, False))
            assert g == e == po.ntt(curve, x, inv, cos)


def test_c_msm_known_discrete_log_2_12():
    curve = po.BLS12_381
    n = 1 << 12
    S, K = ol.random_scalars(curve, n, 1), ol.random_scalars(curve, n, 2)
    B = ol.oracle_g1_mul_gen(curve, K)
    xy, inf = ol.oracle_msm_g1(curve, B, S, 0, 8)
    dot = sum(a * b for a, b in zip(ol.limbs_to_ints(S), ol.limbs_to_ints(K))) % curve.fr.p
    assert ol.limbs_to_point(curve, xy, inf) == po.g1_mul(curve, dot, po.g1_generator(curve))


def test_oracle_timed_variants_agree():
    """bench.py's cpu_baseline legs: forced window width, point-chunked all-core MSM and the multi-threaded NTT arrangement must give the
    same answers as the plain restatements."""
    c = po.BLS12_381
    n = 3000
    k = ol.random_scalars(c, n, 31)
    S = ol.random_scalars(c, n, 32)
    S[0] = 0
    S[1] = ol.ints_to_limbs([1], 4)[0]
    B = ol.oracle_g1_mul_gen(c, k)
    ref, rinf = ol.oracle_msm_g1(c, B, S, algo=0, threads=1)
    for algo, th, co in ((0, 1, 0), (0, 4, 0), (0, 4, 18), (2, 4, 0), (2, 7, 0), (2, 5000, 0)):
        xy, inf, sec = ol.oracle_msm_g1_timed(c, B, S, algo=algo, threads=th, c_override=co)
        assert inf == rinf and (xy == ref).all() and sec > 0
    for curve in (po.BLS12_381, po.BN254):
        x = ol.random_scalars(curve, 1 << 10, 33)
        for inv in (False, True):
            for cos in (False, True):
                want = ol.oracle_ntt(curve, x, inverse=inv, coset=cos, mont=True)
                for th in (1, 3):
                    got, _ = ol.oracle_ntt_timed(curve, x, inverse=inv, coset=cos, threads=th)
                    assert (got == want).all()
