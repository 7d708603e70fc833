"""The lazily reduced 10 x 28-bit scalar field of the NTT passes (openzl_amd/csrc/zl_field28r.h, round 4) against Python integers: pack / unpack,
the carry-free product scan (R' = 2^280), lazy additions, biased subtractions, the top-limb weak reduction and the canonical form -- on the host here,
and on the device (inline-asm scan) under -m gpu."""
import random

import numpy as np
import pytest

from openzl_amd.backend import ZL_BLS12_381, ZL_BN254, hook_fr28_op

R = {ZL_BLS12_381: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, ZL_BN254: 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001}


def _words(vals):
    out = np.zeros((len(vals), 2, 8), dtype=np.uint32)
    for i, (a, b) in enumerate(vals):
        for k in range(8):
            out[i, 0, k] = (a >> (32 * k)) & 0xFFFFFFFF
            out[i, 1, k] = (b >> (32 * k)) & 0xFFFFFFFF
    return out


def _ints(w):
    return [sum(int(w[i, k]) << (32 * k) for k in range(8)) for i in range(w.shape[0])]


def _cases(r, rng, n):
    edge = [0, 1, 2, r - 1, r, r + 1, 2 * r - 1, 2 * r, (1 << 256) - 1, (1 << 252) - 1, 1 << 252, (1 << 28) - 1, 1 << 28, sum(0xFFFFFFF << (28 * i) for i in range(9))]
    vals = [(a, b) for a in edge for b in edge[:8]]
    vals += [(rng.randrange(1 << 256), rng.randrange(1 << 256)) for _ in range(n)]
    vals += [(rng.randrange(r), rng.randrange(r)) for _ in range(n)]
    return vals


def _check(be, curve, bits=28):
    r = R[curve]
    rng = random.Random(bits + curve)
    vals = _cases(r, rng, 400)
    if bits == 29:  # limb boundaries of the nine 29-bit limbs
        edge29 = [(1 << 29) - 1, 1 << 29, sum(0x1FFFFFFF << (29 * i) for i in range(8)), (1 << 232) - 1, 1 << 232]
        vals += [(a, b) for a in edge29 for b in edge29 + [r - 1, (1 << 256) - 1]]
    w = _words(vals)
    rp_inv = pow(1 << (280 if bits == 28 else 261), -1, r)
    op = lambda o, **kw: _ints(hook_fr28_op(be, curve, o, w, bits=bits, **kw))  # noqa: E731
    assert op(0) == [a * b * rp_inv % r for a, b in vals]
    assert op(1) == [(a + b) % r for a, b in vals]
    assert op(3) == [a % r for a, _ in vals]
    assert op(5) == [(2 * a + 2 * b) % r for a, b in vals]  # the weak reduction at the top of its range (2a + 2b < 2^258)
    for j in ((2, 3, 7, 12, 20) if bits == 28 else (2, 3, 5, 6)):  # b < 2^256 < 2.3 r (BLS12-381) / 5.3 r (BN254): j >= 2 resp. 3 covers it
        if (1 << j) < (1 << 256) // r + 2:
            continue
        assert op(2, j=j) == [(a - b) % r for a, b in vals], j
    # the lazy chain: x <- 2x + (x - b^2 R'^-1 + 2^j r) four times (28 bits: bound 3^4 * B(a) + ...; 29 bits: weakly reduced after every step), then one product
    for j in ((2, 5, 11) if bits == 28 else (2, 4, 5)):
        exp = []
        for a, b in vals:
            bb = b * b * rp_inv % r
            x = a
            for _ in range(4):
                x = 3 * x - bb
            exp.append(x * b * rp_inv % r)
        assert op(4, j=j) == exp, j


@pytest.mark.parametrize("curve", [ZL_BLS12_381, ZL_BN254], ids=["bls12_381", "bn254"])
def test_fr28_host(curve):
    _check(None, curve)


@pytest.mark.parametrize("curve", [ZL_BLS12_381, ZL_BN254], ids=["bls12_381", "bn254"])
def test_fr29_host(curve):
    _check(None, curve, bits=29)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [ZL_BLS12_381, ZL_BN254], ids=["bls12_381", "bn254"])
def test_fr28_device(backend, curve):
    _check(backend, curve)


@pytest.mark.gpu
@pytest.mark.parametrize("curve", [ZL_BLS12_381, ZL_BN254], ids=["bls12_381", "bn254"])
def test_fr29_device(backend, curve):
    _check(backend, curve, bits=29)
