"""Worst-case tests of the lazily reduced 14 x 28-bit BLS12-381 Fq (openzl_amd/csrc/zl_field28.h) and of the point formulas built on it.

The field never reduces sums / differences between multiplications: every routine has a contract in units of q.  The contracts are proved
for the formulas at compile time (csrc/zl_bounds.h, static_asserts over an abstract bound domain); here the REAL arithmetic is driven with
operands AT the contract bounds -- values a random MSM never produces -- and compared with Python big integers:
  host path   (-m "not gpu"): the 7 x 56-bit fast path + C++ scans that the host tails use,
  device path (-m gpu):       the single-chain inline-asm product scans the kernels use.
Montgomery radix of the 28-bit field: R' = 2^392.  Round 4: the field arithmetic is driven the same way for the BN254 instance (10 limbs, R' = 2^280; the point
formulas are the same templates, proved once over the abstract bound domain, and gen_params.py asserts that BN254's limits are at least the abstract ones).
"""
import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import po
from openzl_amd.backend import ZL_G1, ZL_G2, hook_fp28_op, hook_point_op

Q = po.BLS12_381.fq.p
RP = 1 << 392
RP_INV = pow(RP, -1, Q)
M28 = (1 << 28) - 1


class _FieldCfg:
    """one instance of zl_field28.h: modulus, limb count, Montgomery radix R' = 2^(28 L), canonical 32-bit words, which test hook"""
    def __init__(self, q: int, L: int, bn254: bool):
        self.Q, self.L, self.bn254 = q, L, bn254
        self.RP = 1 << (28 * L)
        self.RP_INV = pow(self.RP, -1, q)
        self.words = (q.bit_length() + 31) // 32


BLS_FQ = _FieldCfg(po.BLS12_381.fq.p, 14, False)
BN_FQ = _FieldCfg(po.BN254.fq.p, 10, True)   # round 4: BN254 G1 on the same lazily reduced limbs (10 of them, R' = 2^280)


def to_limbs(v: int, L: int = 14) -> np.ndarray:
    """value < 2^(28 L + 4) -> L limbs, limbs 0..L-2 < 2^28, the top limb absorbs the rest (as carry28 leaves it)"""
    out = np.zeros(L, dtype=np.uint32)
    for i in range(L - 1):
        out[i] = (v >> (28 * i)) & M28
    top = v >> (28 * (L - 1))
    assert top < (1 << 32)
    out[L - 1] = top
    return out


def from_limbs(l) -> int:
    return sum(int(x) << (28 * i) for i, x in enumerate(l))


def edge_values(bound: int, rng, count: int, Q: int = Q):
    """values <= bound*q concentrated at the edges: k*q - 1, k*q, k*q + 1 for the top multiples, 0, 1, random residues + (bound-1)*q"""
    vals = [0, 1, Q - 1, Q, bound * Q - 1, bound * Q, max(0, (bound - 1)) * Q + 1]
    vals += [bound * Q - int(rng.integers(1, 1 << 62)) for _ in range(4)]
    while len(vals) < count:
        vals.append(int(rng.integers(0, bound)) * Q + int.from_bytes(rng.bytes(48), "little") % Q)
    return [v for v in vals if 0 <= v <= bound * Q][:count]


def _run_field(be, op, rows, cfg=BLS_FQ):
    arr = np.zeros((len(rows), 4, cfg.L), dtype=np.uint32)
    for i, row in enumerate(rows):
        for j, v in enumerate(row):
            arr[i, j] = to_limbs(v, cfg.L)
    return hook_fp28_op(be, op, arr, bn254=cfg.bn254)


def _check_field(be, cfg=BLS_FQ):
    Q, RP, RP_INV, L, NW = cfg.Q, cfg.RP, cfg.RP_INV, cfg.L, cfg.words
    _ev = edge_values
    edge_values_q = lambda bound, rng_, count: _ev(bound, rng_, count, Q)
    rng = np.random.Generator(np.random.PCG64(28))
    # mul / sqr / muladd: every split of the product budget B(a)*B(b) <= 2500
    for ba, bb in [(1, 1), (2, 2), (8, 2), (10, 10), (16, 16), (50, 50), (2500, 1), (1, 2500), (1250, 2), (100, 25)]:
        A, Bv = edge_values_q(ba, rng, 24), edge_values_q(bb, rng, 24)
        n = min(len(A), len(Bv))
        rows = [(A[i], Bv[(i * 7) % n], 0, 0) for i in range(n)] + [(A[0 if i else -1], Bv[i], 0, 0) for i in range(n)]
        out = _run_field(be, 0, rows, cfg)
        for (a, b, _, _), o in zip(rows, out):
            v = from_limbs(o)
            assert v < 2 * Q and (o <= M28).all() and v % Q == a * b * RP_INV % Q, (ba, bb)
        if ba == bb:
            out = _run_field(be, 1, [(a, 0, 0, 0) for a in A], cfg)
            for a, o in zip(A, out):
                v = from_limbs(o)
                assert v < 2 * Q and (o <= M28).all() and v % Q == a * a * RP_INV % Q, ba
    for (ba, bb, bc, bd) in [(10, 10, 8, 2), (6, 10, 2, 8), (4, 10, 2, 2), (16, 16, 16, 16), (35, 35, 35, 35), (2400, 1, 100, 1), (1, 1250, 1250, 1)]:
        A, Bv, Cv, D = (edge_values_q(b, rng, 20) for b in (ba, bb, bc, bd))
        n = min(map(len, (A, Bv, Cv, D)))
        rows = [(A[i], Bv[(3 * i) % n], Cv[(5 * i) % n], D[(7 * i) % n]) for i in range(n)] + [(A[4], Bv[4], Cv[4], D[4])]
        out = _run_field(be, 2, rows, cfg)
        for (a, b, c, d), o in zip(rows, out):
            v = from_limbs(o)
            assert v < 2 * Q and (o <= M28).all() and v % Q == (a * b + c * d) * RP_INV % Q, (ba, bb, bc, bd)
    # muladd4(a,b,c,d,a,d,c,b): four products, budget 2500 in total (Fq2 components: operands <= 16q)
    for b4 in (16, 25):
        A, Bv, Cv, D = (edge_values_q(b4, rng, 16) for _ in range(4))
        n = min(map(len, (A, Bv, Cv, D)))
        rows = [(A[i], Bv[i], Cv[i], D[i]) for i in range(n)]
        out = _run_field(be, 15, rows, cfg)
        for (a, b, c, d), o in zip(rows, out):
            v = from_limbs(o)
            assert v < 2 * Q and (o <= M28).all() and v % Q == (a * b + c * d + a * d + c * b) * RP_INV % Q
    if True:
        # scan-only operand forms (subk_scan / negk_scan / x3_of: un-carried on the device, carried on the host).  19: a (b - c + 16q) + (16q - d) a with b, c, d carried,
        # c, d < 8q (the bias is one step larger than the carried form's, so that the top limb never wraps); 20: (4q - a) b with a < 2q
        A, Bv, Cv, D = edge_values_q(10, rng, 24), edge_values_q(2, rng, 24), edge_values_q(8, rng, 24), edge_values_q(8, rng, 24)
        Cv = [min(c, 8 * Q - 1) for c in Cv]
        D = [min(d, 8 * Q - 1) for d in D]
        n = min(map(len, (A, Bv, Cv, D)))
        rows = [(A[i], Bv[(3 * i) % n], Cv[(5 * i) % n], D[(7 * i) % n]) for i in range(n)]
        rows += [(10 * Q - 1, 2 * Q - 1, 0, 0), (10 * Q - 1, 0, 8 * Q - 1, 8 * Q - 1), (10 * Q - 1, 2 * Q - 1, 8 * Q - 1, 0), (1, 0, 8 * Q - 1, 8 * Q - 1)]
        out = _run_field(be, 19, rows, cfg)
        for (a, b, c, d), o in zip(rows, out):
            v = from_limbs(o)
            assert v < 2 * Q and (o <= M28).all() and v % Q == (a * (b - c + 16 * Q) + (16 * Q - d) * a) * RP_INV % Q, (a // Q, b // Q, c // Q, d // Q)
        A2 = [min(a, 2 * Q - 1) for a in edge_values_q(2, rng, 24)]
        B2 = edge_values_q(8, rng, len(A2))
        rows = [(a, b, 0, 0) for a, b in zip(A2, B2)] + [(2 * Q - 1, 8 * Q, 0, 0), (0, 8 * Q, 0, 0)]
        out = _run_field(be, 20, rows, cfg)
        for (a, b, _, _), o in zip(rows, out):
            v = from_limbs(o)
            assert v < 2 * Q and (o <= M28).all() and v % Q == (4 * Q - a) * b * RP_INV % Q
        # 21: x3 = a - b - 2c + 6q in one pass; b, c carried values (scan outputs < 2q in the formulas; any carried value is admissible)
        A3, B3, C3 = edge_values_q(2, rng, 24), edge_values_q(2, rng, 24), edge_values_q(2, rng, 24)
        rows = [(a, b, c, 0) for a, b, c in zip(A3, B3, C3)] + [(0, 2 * Q - 1, 2 * Q - 1, 0), (2 * Q - 1, 0, 0, 0), (0, 0, 0, 0)]
        out = _run_field(be, 21, rows, cfg)
        for (a, b, c, _), o in zip(rows, out):
            assert from_limbs(o) == a - b - 2 * c + 6 * Q and (o[:L - 1] <= M28).all()
    # add / dbl: exact integer results, normalised limbs
    A, Bv = edge_values_q(1000, rng, 24), edge_values_q(1000, rng, 24)
    out = _run_field(be, 3, [(a, b, 0, 0) for a, b in zip(A, Bv)], cfg)
    for a, b, o in zip(A, Bv, out):
        assert from_limbs(o) == a + b and (o[:L - 1] <= M28).all()
    out = _run_field(be, 4, [(a, 0, 0, 0) for a in A], cfg)
    for a, o in zip(A, out):
        assert from_limbs(o) == 2 * a and (o[:L - 1] <= M28).all()
    # subk<J>: a - b + 2^J q exactly, for b up to (and at) 2^J q and a from 0 up to a large bound
    for J in range(1, 7):
        Bv = edge_values_q(1 << J, rng, 24)
        A = edge_values_q(64, rng, len(Bv))
        rows = [(a, b, 0, 0) for a, b in zip(A, Bv)] + [(0, b, 0, 0) for b in Bv]
        out = _run_field(be, 4 + J, rows, cfg)
        for (a, b, _, _), o in zip(rows, out):
            assert from_limbs(o) == a - b + (Q << J) and (o[:L - 1] <= M28).all(), J
    # wred (<= 2000 q -> < 4q, same residue), canon, is_zero, ==
    A = edge_values_q(2000, rng, 40) + [k * Q for k in (2, 3, 4, 5, 100, 1999, 2000)] + [k * Q + 1 for k in (3, 4, 1999)] + [k * Q - 1 for k in (1, 4, 2000)]
    out = _run_field(be, 11, [(a, 0, 0, 0) for a in A], cfg)
    for a, o in zip(A, out):
        v = from_limbs(o)
        assert v < 4 * Q and v % Q == a % Q and (o <= M28).all()
    out = _run_field(be, 12, [(a, 0, 0, 0) for a in A], cfg)
    for a, o in zip(A, out):
        assert from_limbs(o) == a % Q
    out = _run_field(be, 13, [(a, 0, 0, 0) for a in A], cfg)
    for a, o in zip(A, out):
        assert int(o[0]) == (1 if a % Q == 0 else 0), a // Q
    Bv = [a + Q * int(rng.integers(0, 3)) if i % 2 else a + 1 for i, a in enumerate(A)]
    out = _run_field(be, 14, [(a, b, 0, 0) for a, b in zip(A, Bv)], cfg)
    for a, b, o in zip(A, Bv, out):
        assert int(o[0]) == (1 if (a - b) % Q == 0 else 0)
    # ABI conversions: canonical 32-bit words -> Montgomery limbs -> canonical words
    canon = [0, 1, Q - 1, Q - 2, int.from_bytes(rng.bytes(48), "little") % Q]
    arr = np.zeros((len(canon), 4, L), dtype=np.uint32)
    for i, v in enumerate(canon):
        arr[i, 0, :NW] = np.array([(v >> (32 * k)) & 0xFFFFFFFF for k in range(NW)], dtype=np.uint32)
    mont = hook_fp28_op(be, 17, arr, bn254=cfg.bn254)
    for v, o in zip(canon, mont):
        assert from_limbs(o) == v * RP % Q
    arr2 = np.zeros((len(canon), 4, L), dtype=np.uint32)
    for i, v in enumerate(canon):
        arr2[i, 0] = to_limbs(v * RP % Q + 3 * Q, L)  # store_canon accepts a lazily reduced value
    back = hook_fp28_op(be, 18, arr2, bn254=cfg.bn254)
    for v, o in zip(canon, back):
        assert sum(int(x) << (32 * k) for k, x in enumerate(o[:NW])) == v


# ---- points ------------------------------------------------------------------------------------------------------------------------
def _mont(v):
    return v * RP % Q


def _lift(res, k):
    """residue + k*q: a representation at a chosen multiple inside the contract"""
    return res + k * Q


def _g1_cases(rng):
    c = po.BLS12_381
    G = po.g1_generator(c)
    pts = [po.g1_mul(c, int(rng.integers(2, 1 << 60)), G) for _ in range(6)]
    cases = []
    for i in range(len(pts)):
        P, Qp = pts[i], pts[(i + 1) % len(pts)]
        cases += [(P, Qp), (P, P), (P, po.g1_neg(c, P)), (None, Qp), (P, None)]
    return cases


def _xyzz_of(P, z, kx, ky, kzz, kzzz, f):
    """XYZZ representation of affine P with zz = z^2, zzz = z^3, coordinates lifted by the given multiples of q; f = field embedding"""
    if P is None:
        return [f.one(kx), f.one(ky), f.zero(), f.zero()]
    zz, zzz = f.mul(z, z), f.mul(f.mul(z, z), z)
    return [f.lift(f.mul(P[0], zz), kx), f.lift(f.mul(P[1], zzz), ky), f.lift(zz, kzz), f.lift(zzz, kzzz)]


class _Fq:
    W = 14

    @staticmethod
    def mul(a, b):
        return a * b % Q

    @staticmethod
    def lift(v, k):
        return to_limbs(_lift(_mont(v), k))

    @staticmethod
    def one(k):
        return to_limbs(_lift(_mont(1), k))

    @staticmethod
    def zero():
        return np.zeros(14, dtype=np.uint32)

    @staticmethod
    def rand(rng):
        return int.from_bytes(rng.bytes(48), "little") % (Q - 1) + 1

    @staticmethod
    def decode(limbs):
        return from_limbs(limbs) * RP_INV % Q

    @staticmethod
    def inv(a):
        return pow(a, -1, Q)

    @staticmethod
    def is_zero(a):
        return a == 0

    @staticmethod
    def maxval(limbs):
        return from_limbs(limbs)


class _Fq2:
    W = 28

    @staticmethod
    def mul(a, b):
        return po.f2_mul(Q, a, b)

    @staticmethod
    def lift(v, k):
        return np.concatenate([to_limbs(_lift(_mont(v[0]), k)), to_limbs(_lift(_mont(v[1]), k))])

    @staticmethod
    def one(k):
        return _Fq2.lift((1, 0), k)

    @staticmethod
    def zero():
        return np.zeros(28, dtype=np.uint32)

    @staticmethod
    def rand(rng):
        return (_Fq.rand(rng), _Fq.rand(rng))

    @staticmethod
    def decode(limbs):
        return (from_limbs(limbs[:14]) * RP_INV % Q, from_limbs(limbs[14:]) * RP_INV % Q)

    @staticmethod
    def inv(a):
        return po.f2_inv(Q, a)

    @staticmethod
    def is_zero(a):
        return a == (0, 0)

    @staticmethod
    def maxval(limbs):
        return max(from_limbs(limbs[:14]), from_limbs(limbs[14:]))


def _affine_of(f, out):
    """decode an XYZZ result (raw limbs) to an affine big-int point; checks the closure contract (coordinates < 8q, limbs normalised)"""
    W = f.W
    for k in range(4):
        assert f.maxval(out[k]) < 8 * Q
        for h in range(W // 14):
            assert (out[k][14 * h: 14 * h + 13] <= M28).all()
    zz = f.decode(out[2])
    if f.is_zero(zz):
        return None
    zzz = f.decode(out[3])
    return (f.mul(f.decode(out[0]), f.inv(zz)), f.mul(f.decode(out[1]), f.inv(zzz)))


def _check_points(be, group, hot):
    rng = np.random.Generator(np.random.PCG64(7 + group))
    c = po.BLS12_381
    if group == ZL_G1:
        f, add, neg, G, mulp = _Fq, po.g1_add, po.g1_neg, po.g1_generator(c), po.g1_mul
    else:
        f, add, G, mulp = _Fq2, po.g2_add, po.g2_generator(c), po.g2_mul
        neg = lambda cc, P: None if P is None else (P[0], ((-P[1][0]) % Q, (-P[1][1]) % Q))  # noqa: E731
    pts = [mulp(c, int(rng.integers(2, 1 << 40)), G) for _ in range(4)]
    pairs = []
    for i in range(len(pts)):
        P, Qp = pts[i], pts[(i + 1) % len(pts)]
        pairs += [(P, Qp), (P, P), (P, neg(c, P)), (None, Qp), (P, None), (None, None)]
    W = f.W
    # coordinate lifts: all at the contract maximum (value in [7q, 8q)), all canonical, mixed
    lifts = [(7, 7, 7, 7), (0, 0, 0, 0), (7, 0, 7, 0), (3, 7, 1, 5)]
    for op in (0, 1, 2, 3, 4, 5, 6):
        rows, expect = [], []
        for (P, Qp) in pairs:
            for lp in lifts:
                zp, zq = f.rand(rng), f.rand(rng)
                if op in (0, 1):  # mixed: q affine, canonical or < 2q (zz ignored); q must not be infinity
                    if Qp is None:
                        continue
                    for kq in (0, 1):
                        pp = _xyzz_of(P, zp, *lp, f)
                        qq = [f.lift(Qp[0], kq), f.lift(Qp[1], kq), f.one(0), f.one(0)]
                        rows.append(pp + qq)
                        expect.append(add(c, P, Qp if op == 0 else neg(c, Qp)))
                elif op == 2:
                    rows.append(_xyzz_of(P, zp, *lp, f) + _xyzz_of(Qp, zq, *lp[::-1], f))
                    expect.append(add(c, P, Qp))
                elif op == 3:
                    rows.append(_xyzz_of(P, zp, *lp, f) + _xyzz_of(None, 1, 0, 0, 0, 0, f))
                    expect.append(add(c, P, P))
                elif op == 4:  # dbl_affine: affine coordinates up to 8q, never infinity
                    if P is None:
                        continue
                    rows.append([f.lift(P[0], lp[0]), f.lift(P[1], lp[1]), f.one(0), f.one(0)] + _xyzz_of(None, 1, 0, 0, 0, 0, f))
                    expect.append(add(c, P, P))
                elif op == 5:
                    rows.append(_xyzz_of(P, zp, *lp, f) + _xyzz_of(None, 1, 0, 0, 0, 0, f))
                    expect.append(neg(c, P))
                else:
                    rows.append(_xyzz_of(P, zp, *lp, f) + _xyzz_of(None, 1, 0, 0, 0, 0, f))
                    expect.append(P)
        arr = np.stack([np.stack(r) for r in rows]).astype(np.uint32)
        out = hook_point_op(be, group, hot, op, arr)
        for o, e, r in zip(out, expect, rows):
            got = _affine_of(f, o)
            assert got == e, (group, hot, op)
            if op == 6 and e is not None:  # to_affine returns canonical coordinates
                assert f.maxval(o[0]) < Q and f.maxval(o[1]) < Q


@pytest.mark.parametrize("cfg", [BLS_FQ, BN_FQ], ids=["bls12_381", "bn254"])
def test_fp28_contract_edges_host(cfg):
    _check_field(None, cfg)


@pytest.mark.parametrize("group,hot", [(ZL_G1, False), (ZL_G2, False), (ZL_G2, True)], ids=["g1", "g2-called", "g2-inlined"])
def test_point_formulas_at_bounds_host(group, hot):
    _check_points(None, group, hot)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [BLS_FQ, BN_FQ], ids=["bls12_381", "bn254"])
def test_fp28_contract_edges_device(backend, cfg):
    _check_field(backend, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("group,hot", [(ZL_G1, False), (ZL_G2, False), (ZL_G2, True)], ids=["g1", "g2-called", "g2-inlined"])
def test_point_formulas_at_bounds_device(backend, group, hot):
    _check_points(backend, group, hot)
