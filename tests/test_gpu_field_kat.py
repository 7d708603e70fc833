"""The one reference-held known-answer vector the HIP path can meet directly: the Poseidon permutation of [3, 1, 2]
over BLS12-381 Fr, computed ON THE DEVICE with the device Montgomery arithmetic (to_mont, add, mul, sqr, Fermat inversion for the MDS
matrix, from_mont -- thousands of device multiplications), must equal the numbers of
/root/reference/openzl-tutorials/src/poseidon.rs:383-400 (= plugins/arkworks/src/poseidon/permutation_hardcoded_test/width3), which
tests/golden/ref_poseidon_fixtures.json holds as data (extracted by tests/golden/make_ref_fixtures.py)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd.backend import ZL_BLS12_381, ZL_BN254, poseidon_permute, hook_poseidon_permute_dev, hook_poseidon_permute_dev28r

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_poseidon_fixtures.json")


def test_device_poseidon_permutation_matches_reference_fixture(backend):
    fx = json.load(open(GOLD))["permutation_width3"]
    inp = [int(v) for v in fx["input"]]
    exp = [int(v) for v in fx["output"]]
    assert inp == [3, 1, 2]
    got = hook_poseidon_permute_dev(backend, ZL_BLS12_381, ol.ints_to_limbs(inp, 4))
    assert ol.limbs_to_ints(got) == exp
    # SURVEY.md §8c.1 quotes the same three numbers
    assert exp[0] == 1808609226548932412441401219270714120272118151392880709881321306315053574086


@pytest.mark.parametrize("curve,cid", [(po.BLS12_381, ZL_BLS12_381), (po.BN254, ZL_BN254)], ids=["bls12_381", "bn254"])
def test_device_poseidon_matches_oracle_and_host_mirror(backend, curve, cid):
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(4):
        st = [int.from_bytes(rng.bytes(40), "little") % curve.fr.p for _ in range(3)]
        got = ol.limbs_to_ints(hook_poseidon_permute_dev(backend, cid, ol.ints_to_limbs(st, 4)))
        assert got == po.poseidon_permute(curve.fr, st)
        assert got == ol.limbs_to_ints(poseidon_permute(cid, ol.ints_to_limbs(st, 4)))
    edge = [0, curve.fr.p - 1, 1]
    assert ol.limbs_to_ints(hook_poseidon_permute_dev(backend, cid, ol.ints_to_limbs(edge, 4))) == po.poseidon_permute(curve.fr, edge)


def test_device_poseidon_on_the_ntt_multiplier_matches_reference_fixture(backend):
    """since round 4 the NTT passes multiply on the lazily reduced 10 x 28-bit Fr (zl_field28r.h, mul28r_asm), which the hook above
    does not touch.  The reference's [3, 1, 2] vector (/root/reference/openzl-tutorials/src/poseidon.rs:364-405) through THAT multiplier: conversions by
    R'^2, 63 rounds of lazy additions and products, nine Fermat inversions for the Cauchy matrix, one canon at the end."""
    fx = json.load(open(GOLD))["permutation_width3"]
    exp = [int(v) for v in fx["output"]]
    got = hook_poseidon_permute_dev28r(backend, ZL_BLS12_381, ol.ints_to_limbs([int(v) for v in fx["input"]], 4))
    assert ol.limbs_to_ints(got) == exp


@pytest.mark.parametrize("curve,cid", [(po.BLS12_381, ZL_BLS12_381), (po.BN254, ZL_BN254)], ids=["bls12_381", "bn254"])
def test_device_poseidon_on_the_ntt_multiplier_matches_oracle(backend, curve, cid):
    rng = np.random.Generator(np.random.PCG64(6))
    cases = [[int.from_bytes(rng.bytes(40), "little") % curve.fr.p for _ in range(3)] for _ in range(4)] + [[0, curve.fr.p - 1, 1], [curve.fr.p - 1] * 3]
    for st in cases:
        got = ol.limbs_to_ints(hook_poseidon_permute_dev28r(backend, cid, ol.ints_to_limbs(st, 4)))
        assert got == po.poseidon_permute(curve.fr, st)
        assert got == ol.limbs_to_ints(hook_poseidon_permute_dev(backend, cid, ol.ints_to_limbs(st, 4)))
