"""Pure-Python big-int oracle for the OpenZL arkworks MSM / NTT / Groth16 hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``openzl_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.

Parity status: **unpinned at the MSM/NTT/Groth16 boundary** (the reference holds no golden
vector for those) and **pinned for BLS12-381 Fr arithmetic** against the
reference's own fixtures (tests/golden/ref_poseidon_fixtures.json, extracted from
/root/reference/plugins/arkworks/src/poseidon/{mds_hardcoded_tests,parameters_hardcoded_test,
permutation_hardcoded_test}).  MSM / NTT results are additionally pinned by mathematical
uniqueness: this file computes them from the *definitions* (sum of scalar multiples, DFT sum),
not from the fast algorithms, so that the C restatement (oracle/zl_oracle.c) and the HIP path
can both be checked against something independent of Pippenger / Cooley-Tukey.

The algorithms restated here live in third-party crates that are NOT vendored under
/root/reference (plugins/arkworks/Cargo.toml:113-146): ark-ec 0.3.0 (VariableBaseMSM),
ark-poly 0.3.0 (Radix2EvaluationDomain), ark-ff 0.3.0 (Fp256/Fp384 Montgomery),
ark-groth16 0.3.0 (R1CStoQAP::witness_map, create_proof_with_assignment).  The reference
call sites are plugins/arkworks/src/groth16.rs:438 (setup) and :454 (prove).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------------------------
# Field / curve parameters (re-verified by tests/test_oracle_py.py)
# --------------------------------------------------------------------------------------------


@dataclass(frozen=True)
class FieldParams:
    name: str
    p: int
    limbs64: int  # number of 64-bit limbs (ark-ff Fp256 -> 4, Fp384 -> 6)

    @property
    def R(self) -> int:  # Montgomery radix (ark-ff: 2^(64*limbs))
        return 1 << (64 * self.limbs64)

    @property
    def bits(self) -> int:
        return self.p.bit_length()

    def to_mont(self, x: int) -> int:
        return (x * self.R) % self.p

    def from_mont(self, x: int) -> int:
        return (x * pow(self.R, -1, self.p)) % self.p

    @property
    def inv64(self) -> int:  # -p^{-1} mod 2^64
        return (-pow(self.p, -1, 1 << 64)) % (1 << 64)

    @property
    def inv32(self) -> int:
        return (-pow(self.p, -1, 1 << 32)) % (1 << 32)


BLS12_381_FQ = FieldParams(
    "bls12_381_fq",
    0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    6,
)
BLS12_381_FR = FieldParams(
    "bls12_381_fr",
    0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    4,
)
BN254_FQ = FieldParams(
    "bn254_fq",
    0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
    4,
)
BN254_FR = FieldParams(
    "bn254_fr",
    0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001,
    4,
)


@dataclass(frozen=True)
class CurveParams:
    name: str
    cid: int  # zl_curve_t value (include/zl_backend.h)
    fq: FieldParams
    fr: FieldParams
    b: int  # y^2 = x^3 + b
    gx: int
    gy: int
    fr_generator: int  # multiplicative generator of Fr (ark: GENERATOR)
    two_adicity: int
    # G2: y^2 = x^3 + b2 over Fq2 = Fq[u]/(u^2+1)
    b2: Tuple[int, int] = (0, 0)
    g2x: Tuple[int, int] = (0, 0)
    g2y: Tuple[int, int] = (0, 0)

    @property
    def two_adic_root(self) -> int:  # ark: TWO_ADIC_ROOT_OF_UNITY = g^((r-1)/2^s)
        r = self.fr.p
        return pow(self.fr_generator, (r - 1) >> self.two_adicity, r)


BLS12_381 = CurveParams(
    "bls12_381",
    1,
    BLS12_381_FQ,
    BLS12_381_FR,
    4,
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    7,
    32,
    b2=(4, 4),
    g2x=(
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    g2y=(
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)
BN254 = CurveParams(
    "bn254",
    2,
    BN254_FQ,
    BN254_FR,
    3,
    1,
    2,
    5,
    28,
    b2=(
        19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690,
    ),
    g2x=(
        10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634,
    ),
    g2y=(
        8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531,
    ),
)

CURVES = {"bls12_381": BLS12_381, "bn254": BN254}

# --------------------------------------------------------------------------------------------
# G1 affine arithmetic straight from the group law (None = point at infinity)
# --------------------------------------------------------------------------------------------

Point = Optional[Tuple[int, int]]


def g1_is_on_curve(c: CurveParams, P: Point) -> bool:
    if P is None:
        return True
    x, y = P
    p = c.fq.p
    return (y * y - x * x * x - c.b) % p == 0


def g1_neg(c: CurveParams, P: Point) -> Point:
    if P is None:
        return None
    return (P[0], (-P[1]) % c.fq.p)


def g1_add(c: CurveParams, P: Point, Q: Point) -> Point:
    p = c.fq.p
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def g1_mul(c: CurveParams, k: int, P: Point) -> Point:
    """Double-and-add (definition of scalar multiplication)."""
    k %= c.fr.p
    acc: Point = None
    add = P
    while k:
        if k & 1:
            acc = g1_add(c, acc, add)
        add = g1_add(c, add, add)
        k >>= 1
    return acc


def g1_generator(c: CurveParams) -> Point:
    return (c.gx, c.gy)


def msm_naive(c: CurveParams, scalars: Sequence[int], points: Sequence[Point]) -> Point:
    """sum_i scalars[i] * points[i] by the definition; the uniqueness anchor for every MSM."""
    acc: Point = None
    for s, P in zip(scalars, points):
        acc = g1_add(c, acc, g1_mul(c, s, P))
    return acc


def ark_window_bits(n: int) -> int:
    """ark-ec 0.3.0 VariableBaseMSM: c = 3 if n < 32 else ln_without_floats(n) + 2 where
    ln_without_floats(a) = log2(a) * 69 / 100 with ark_std::log2 = ceil(log2)."""
    if n < 32:
        return 3
    ceil_log2 = (n - 1).bit_length()
    return ceil_log2 * 69 // 100 + 2


def msm_pippenger_ark(c: CurveParams, scalars: Sequence[int], points: Sequence[Point]) -> Point:
    """Restatement of ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul on affine
    big-int points; used on small inputs to check that the arkworks schedule (unsigned windows,
    zero skip, unit-scalar shortcut, running-sum reduce, Horner combine) equals msm_naive."""
    n = min(len(scalars), len(points))
    cbits = ark_window_bits(n)
    num_bits = c.fr.bits
    window_sums: List[Point] = []
    for w_start in range(0, num_bits, cbits):
        res: Point = None
        buckets: List[Point] = [None] * ((1 << cbits) - 1)
        for s, P in zip(scalars[:n], points[:n]):
            if s == 0:
                continue
            if s == 1:
                if w_start == 0:
                    res = g1_add(c, res, P)
                continue
            d = (s >> w_start) & ((1 << cbits) - 1)
            if d != 0:
                buckets[d - 1] = g1_add(c, buckets[d - 1], P)
        run: Point = None
        for b in reversed(buckets):
            run = g1_add(c, run, b)
            res = g1_add(c, res, run)
        window_sums.append(res)
    lowest = window_sums[0]
    total: Point = None
    for ws in reversed(window_sums[1:]):
        total = g1_add(c, total, ws)
        for _ in range(cbits):
            total = g1_add(c, total, total)
    return g1_add(c, lowest, total)


# --------------------------------------------------------------------------------------------
# Fq2 / G2 (row f1) -- Fq2 = Fq[u]/(u^2+1)
# --------------------------------------------------------------------------------------------

F2 = Tuple[int, int]
Point2 = Optional[Tuple[F2, F2]]


def f2_add(p: int, a: F2, b: F2) -> F2:
    return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)


def f2_sub(p: int, a: F2, b: F2) -> F2:
    return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)


def f2_mul(p: int, a: F2, b: F2) -> F2:
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(p: int, a: F2) -> F2:
    n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
    return (a[0] * n % p, (-a[1]) * n % p)


def g2_is_on_curve(c: CurveParams, P: Point2) -> bool:
    if P is None:
        return True
    p = c.fq.p
    x, y = P
    lhs = f2_mul(p, y, y)
    rhs = f2_add(p, f2_mul(p, f2_mul(p, x, x), x), c.b2)
    return lhs == rhs


def g2_add(c: CurveParams, P: Point2, Q: Point2) -> Point2:
    p = c.fq.p
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if f2_add(p, y1, y2) == (0, 0):
            return None
        xx = f2_mul(p, x1, x1)
        num = f2_add(p, f2_add(p, xx, xx), xx)
        lam = f2_mul(p, num, f2_inv(p, f2_add(p, y1, y1)))
    else:
        lam = f2_mul(p, f2_sub(p, y2, y1), f2_inv(p, f2_sub(p, x2, x1)))
    x3 = f2_sub(p, f2_sub(p, f2_mul(p, lam, lam), x1), x2)
    y3 = f2_sub(p, f2_mul(p, lam, f2_sub(p, x1, x3)), y1)
    return (x3, y3)


def g2_mul(c: CurveParams, k: int, P: Point2) -> Point2:
    k %= c.fr.p
    acc: Point2 = None
    add = P
    while k:
        if k & 1:
            acc = g2_add(c, acc, add)
        add = g2_add(c, add, add)
        k >>= 1
    return acc


def g2_generator(c: CurveParams) -> Point2:
    return (c.g2x, c.g2y)


def msm_g2_naive(c: CurveParams, scalars: Sequence[int], points: Sequence[Point2]) -> Point2:
    acc: Point2 = None
    for s, P in zip(scalars, points):
        acc = g2_add(c, acc, g2_mul(c, s, P))
    return acc


# --------------------------------------------------------------------------------------------
# NTT over Fr -- ark-poly 0.3.0 Radix2EvaluationDomain semantics (natural order in and out)
# --------------------------------------------------------------------------------------------


def domain_root(c: CurveParams, log_n: int) -> int:
    """group_gen = TWO_ADIC_ROOT_OF_UNITY ^ (2^(TWO_ADICITY - log_n))."""
    assert 0 <= log_n <= c.two_adicity
    return pow(c.two_adic_root, 1 << (c.two_adicity - log_n), c.fr.p)


def dft_naive(c: CurveParams, x: Sequence[int], inverse: bool = False, coset: bool = False) -> List[int]:
    """X_k = sum_j x_j w^{jk} by the definition (O(n^2)); inverse multiplies by n^{-1} and uses
    w^{-1}; coset variants pre-multiply x_j by g^j (forward) / post-multiply by g^{-j} (inverse)."""
    r = c.fr.p
    n = len(x)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = domain_root(c, log_n)
    g = c.fr_generator
    x = [v % r for v in x]
    if inverse:
        w = pow(w, -1, r)
    elif coset:
        x = [v * pow(g, j, r) % r for j, v in enumerate(x)]
    out = []
    for k in range(n):
        wk = pow(w, k, r)
        acc = 0
        wjk = 1
        for j in range(n):
            acc = (acc + x[j] * wjk) % r
            wjk = wjk * wk % r
        out.append(acc)
    if inverse:
        ninv = pow(n, -1, r)
        out = [v * ninv % r for v in out]
        if coset:
            ginv = pow(g, -1, r)
            out = [v * pow(ginv, j, r) % r for j, v in enumerate(out)]
    return out


def ntt(c: CurveParams, x: Sequence[int], inverse: bool = False, coset: bool = False) -> List[int]:
    """Iterative radix-2 NTT with the same semantics as dft_naive (fast path for tests)."""
    r = c.fr.p
    n = len(x)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    w = domain_root(c, log_n)
    g = c.fr_generator
    a = [v % r for v in x]
    if inverse:
        w = pow(w, -1, r)
    elif coset:
        gp = 1
        for j in range(n):
            a[j] = a[j] * gp % r
            gp = gp * g % r
    # bit reversal then DIT
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, r)
        for i in range(0, n, length):
            wj = 1
            for k in range(length // 2):
                u = a[i + k]
                v = a[i + k + length // 2] * wj % r
                a[i + k] = (u + v) % r
                a[i + k + length // 2] = (u - v) % r
                wj = wj * wl % r
        length <<= 1
    if inverse:
        ninv = pow(n, -1, r)
        a = [v * ninv % r for v in a]
        if coset:
            ginv = pow(g, -1, r)
            gp = 1
            for j in range(n):
                a[j] = a[j] * gp % r
                gp = gp * ginv % r
    return a


# --------------------------------------------------------------------------------------------
# Poseidon over Fr (config 5; pinned by the reference's fixtures)
# --------------------------------------------------------------------------------------------


class GrainLFSR:
    """Restates openzl-crypto/src/poseidon/lfsr.rs:14-100 (80-bit Grain LFSR, GKRRS19 App. A)."""

    SIZE = 80

    def __init__(self, seed: Sequence[Tuple[int, int]]):
        self.state = [False] * self.SIZE
        self.head = 0
        for n, bits in seed:  # append_seed_bits, lfsr.rs:44-50 (MSB first)
            for i in reversed(range(n)):
                self._set_next(bool((bits >> i) & 1))
        for _ in range(self.SIZE * 2):  # skip_updates(160), lfsr.rs:40
            self._update()

    def _set_next(self, b: bool) -> bool:
        self.state[self.head] = b
        self.head = (self.head + 1) % self.SIZE
        return b

    def _bit(self, i: int) -> bool:
        return self.state[(i + self.head) % self.SIZE]

    def _update(self) -> bool:  # lfsr.rs:76-80
        return self._set_next(
            self._bit(62) ^ self._bit(51) ^ self._bit(38) ^ self._bit(23) ^ self._bit(13) ^ self._bit(0)
        )

    def next_bit(self) -> bool:  # Iterator::next, lfsr.rs:86-93 (self-shrinking)
        bit = self._update()
        while not bit:
            self._update()
            bit = self._update()
        return self._update()


def poseidon_round_constants(field: FieldParams, width: int, rf: int, rp: int) -> List[int]:
    """openzl-crypto/src/poseidon/round_constants.rs:10-59: MODULUS_BITS bits big-endian per
    candidate, rejection if >= modulus."""
    lfsr = GrainLFSR(
        [(2, 1), (4, 0), (12, field.bits), (12, width), (10, rf), (10, rp), (30, (1 << 30) - 1)]
    )
    out = []
    while len(out) < width * (rf + rp):
        v = 0
        for _ in range(field.bits):
            v = (v << 1) | int(lfsr.next_bit())
        if v < field.p:
            out.append(v)
    return out


def poseidon_mds(field: FieldParams, t: int) -> List[List[int]]:
    """openzl-crypto/src/poseidon/mds.rs:84-102: M[i][j] = (i + (t + j))^{-1}."""
    return [[pow(i + t + j, -1, field.p) for j in range(t)] for i in range(t)]


def poseidon_permute(field: FieldParams, state: Sequence[int], rf: int = 8, rp: int = 55) -> List[int]:
    """openzl-tutorials/src/poseidon.rs:165-222: every round = add keys -> S-box (x^5; all lanes
    in full rounds, lane 0 in partial rounds) -> MDS multiply."""
    p = field.p
    t = len(state)
    keys = poseidon_round_constants(field, t, rf, rp)
    mds = poseidon_mds(field, t)
    s = [v % p for v in state]
    half = rf // 2
    for rnd in range(rf + rp):
        k = keys[rnd * t:(rnd + 1) * t]
        s = [(v + kk) % p for v, kk in zip(s, k)]
        if rnd < half or rnd >= half + rp:
            s = [pow(v, 5, p) for v in s]
        else:
            s[0] = pow(s[0], 5, p)
        s = [sum(mds[i][j] * s[j] for j in range(t)) % p for i in range(t)]
    return s


# --------------------------------------------------------------------------------------------
# Deterministic inputs shared by tests / bench (SplitMix64)
# --------------------------------------------------------------------------------------------

MASK64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)


def sample_fr(field: FieldParams, rng: SplitMix64) -> int:
    """4 x u64 little-endian, masked to MODULUS_BITS, rejection until < modulus."""
    while True:
        v = 0
        for i in range(field.limbs64):
            v |= rng.next() << (64 * i)
        v &= (1 << field.bits) - 1
        if v < field.p:
            return v


def to_limbs(x: int, nlimbs64: int) -> List[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(nlimbs64)]


def from_limbs(limbs: Sequence[int]) -> int:
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    return v


# --------------------------------------------------------------------------------------------
# R1CS + Groth16 (config 5).  Restates ark-groth16 0.3.0 (third-party, not vendored; pinned
# plugins/arkworks/Cargo.toml:113; reached from plugins/arkworks/src/groth16.rs:438 `circuit_specific_setup`
# and :454 `prove`): R1CStoQAP::witness_map, generate_parameters (with an explicit, KNOWN trapdoor so that
# every proof element can also be recomputed in the exponent) and
# create_proof_with_assignment.  Circuit synthesis is CPU-side in the reference too
# (plugins/arkworks/src/constraint/mod.rs:64-108,179-197).
# --------------------------------------------------------------------------------------------


class LC:
    """linear combination sum coeff*var; variables are ('i', k) instance (k = 0 is the constant ONE) or ('w', k)."""

    __slots__ = ("t",)

    def __init__(self, terms=None):
        self.t = dict(terms or {})

    @staticmethod
    def const(v: int) -> "LC":
        return LC({("i", 0): v}) if v else LC()

    def is_const(self) -> bool:
        return all(k == ("i", 0) for k in self.t)

    def add(self, o: "LC", p: int) -> "LC":
        r = dict(self.t)
        for k, v in o.t.items():
            nv = (r.get(k, 0) + v) % p
            if nv:
                r[k] = nv
            else:
                r.pop(k, None)
        return LC(r)

    def scale(self, c: int, p: int) -> "LC":
        c %= p
        return LC({k: v * c % p for k, v in self.t.items()}) if c else LC()


class R1CS:
    """A*z o B*z = C*z with z = (1, public..., witness...) -- the layout ark-relations hands to ark-groth16."""

    def __init__(self, field: FieldParams):
        self.f = field
        self.pub = [1]  # instance assignment, index 0 = ONE
        self.wit: List[int] = []
        self.A: List[LC] = []
        self.B: List[LC] = []
        self.C: List[LC] = []

    def new_public(self, value: int) -> LC:
        self.pub.append(value % self.f.p)
        return LC({("i", len(self.pub) - 1): 1})

    def new_witness(self, value: int) -> LC:
        self.wit.append(value % self.f.p)
        return LC({("w", len(self.wit) - 1): 1})

    def value(self, lc: LC) -> int:
        p = self.f.p
        acc = 0
        for (kind, k), c in lc.t.items():
            acc += c * (self.pub[k] if kind == "i" else self.wit[k])
        return acc % p

    def enforce(self, a: LC, b: LC, c: LC):
        self.A.append(a)
        self.B.append(b)
        self.C.append(c)

    def mul(self, a: LC, b: LC) -> LC:
        """allocate out = a*b (one constraint); constants fold without a constraint"""
        p = self.f.p
        if a.is_const():
            return b.scale(self.value(a), p)
        if b.is_const():
            return a.scale(self.value(b), p)
        out = self.new_witness(self.value(a) * self.value(b))
        self.enforce(a, b, out)
        return out

    # -- shape as arkworks reports it
    @property
    def n_constraints(self) -> int:
        return len(self.A)

    @property
    def n_instance(self) -> int:
        return len(self.pub)

    @property
    def n_witness(self) -> int:
        return len(self.wit)

    def var_index(self, key) -> int:
        return key[1] if key[0] == "i" else self.n_instance + key[1]

    def assignment(self) -> List[int]:
        return list(self.pub) + list(self.wit)

    def is_satisfied(self) -> bool:
        p = self.f.p
        return all(self.value(a) * self.value(b) % p == self.value(c) for a, b, c in zip(self.A, self.B, self.C))

    def csr(self, which: str):
        rows = {"A": self.A, "B": self.B, "C": self.C}[which]
        ptr, col, val = [0], [], []
        for lc in rows:
            for k in sorted(lc.t, key=self.var_index):
                col.append(self.var_index(k))
                val.append(lc.t[k])
            ptr.append(len(col))
        return ptr, col, val

    def domain_log(self) -> int:
        need = self.n_constraints + self.n_instance
        return max(1, (need - 1).bit_length())


def poseidon_hash_gadget(cs: R1CS, x: LC, y: LC, keys, mds, rf: int = 8, rp: int = 55) -> LC:
    """In-circuit Poseidon arity-2 hash: state (2^arity - 1, x, y), tutorial/hasher schedule
    (openzl-crypto/src/poseidon/hash.rs:93-104,123-135, mod.rs:229-282; plugin ops
    plugins/arkworks/src/poseidon/mod.rs:225-298: add/add_const/mul_const are linear (0 constraints),
    apply_sbox = x^5 = 3 multiplication constraints).  Returns lane 0."""
    p = cs.f.p
    t = 3
    state = [LC.const(3), x, y]
    half = rf // 2
    for rnd in range(rf + rp):
        k = keys[rnd * t:(rnd + 1) * t]
        state = [s.add(LC.const(kk), p) for s, kk in zip(state, k)]
        lanes = range(t) if (rnd < half or rnd >= half + rp) else range(1)
        for i in lanes:
            v = state[i]
            x2 = cs.mul(v, v)
            x4 = cs.mul(x2, x2)
            state[i] = cs.mul(x4, v)
        nxt = []
        for i in range(t):
            acc = LC()
            for j in range(t):
                acc = acc.add(state[j].scale(mds[i][j], p), p)
            nxt.append(acc)
        state = nxt
    return state[0]


def poseidon_chain_circuit(field: FieldParams, k: int, x0: int = 1, x1: int = 2) -> R1CS:
    """config 5: h_1 = H(x0, x1), h_{j+1} = H(h_j, x1); public input = h_k."""
    cs = R1CS(field)
    keys = poseidon_round_constants(field, 3, 8, 55)
    mds = poseidon_mds(field, 3)
    expected = x0 % field.p
    for _ in range(k):
        expected = poseidon_permute(field, [3, expected, x1])[0]
    out_pub = cs.new_public(expected)  # publics are allocated first (instance block precedes witnesses)
    a = cs.new_witness(x0)
    b = cs.new_witness(x1)
    h = a
    for _ in range(k):
        h = poseidon_hash_gadget(cs, h, b, keys, mds)
    cs.enforce(h, LC.const(1), out_pub)
    assert cs.value(h) == expected
    return cs


def qap_witness_map(c: CurveParams, cs: R1CS) -> List[int]:
    """ark-groth16 0.3.0 R1CStoQAP::witness_map: h coefficients, N of them."""
    r = c.fr.p
    n = 1 << cs.domain_log()
    z = cs.assignment()
    a = [0] * n
    b = [0] * n
    cc = [0] * n
    for i in range(cs.n_constraints):
        a[i] = cs.value(cs.A[i])
        b[i] = cs.value(cs.B[i])
        cc[i] = cs.value(cs.C[i])
    for j in range(cs.n_instance):
        a[cs.n_constraints + j] = z[j]
    a = ntt(c, ntt(c, a, inverse=True), coset=True)
    b = ntt(c, ntt(c, b, inverse=True), coset=True)
    cc = ntt(c, ntt(c, cc, inverse=True), coset=True)
    zinv = pow(pow(c.fr_generator, n, r) - 1, -1, r)  # vanishing polynomial on the coset is the constant g^n - 1
    ab = [((x * y - w) % r) * zinv % r for x, y, w in zip(a, b, cc)]
    return ntt(c, ab, inverse=True, coset=True)


def lagrange_at(c: CurveParams, log_n: int, tau: int) -> List[int]:
    """evaluate_all_lagrange_coefficients(tau) for tau outside the domain: L_j = Z(tau)/n * w^j / (tau - w^j)"""
    r = c.fr.p
    n = 1 << log_n
    w = domain_root(c, log_n)
    zt = (pow(tau, n, r) - 1) % r
    ninv = pow(n, -1, r)
    out = []
    wj = 1
    for _ in range(n):
        out.append(zt * ninv % r * wj % r * pow((tau - wj) % r, -1, r) % r)
        wj = wj * w % r
    return out


@dataclass
class Groth16Trapdoor:
    alpha: int
    beta: int
    gamma: int
    delta: int
    tau: int


def groth16_setup_exponents(c: CurveParams, cs: R1CS, td: Groth16Trapdoor):
    """ark-groth16 0.3.0 generate_parameters in the exponent: discrete logs (w.r.t. the G1 / G2 generators) of
    every proving-key element.  Returns dict of lists / scalars (all mod r)."""
    r = c.fr.p
    log_n = cs.domain_log()
    n = 1 << log_n
    L = lagrange_at(c, log_n, td.tau)
    nv = cs.n_instance + cs.n_witness
    u = [0] * nv
    v = [0] * nv
    w = [0] * nv
    for j in range(cs.n_instance):
        u[j] = L[cs.n_constraints + j]
    for row in range(cs.n_constraints):
        for k, coef in cs.A[row].t.items():
            i = cs.var_index(k)
            u[i] = (u[i] + coef * L[row]) % r
        for k, coef in cs.B[row].t.items():
            i = cs.var_index(k)
            v[i] = (v[i] + coef * L[row]) % r
        for k, coef in cs.C[row].t.items():
            i = cs.var_index(k)
            w[i] = (w[i] + coef * L[row]) % r
    zt = (pow(td.tau, n, r) - 1) % r
    dinv = pow(td.delta, -1, r)
    ginv = pow(td.gamma, -1, r)
    comb = [(td.beta * u[i] + td.alpha * v[i] + w[i]) % r for i in range(nv)]
    h = []
    tp = 1
    for _ in range(n - 1):
        h.append(zt * dinv % r * tp % r)
        tp = tp * td.tau % r
    return {
        "a_query": u, "b_query": v,
        "h_query": h,
        "l_query": [comb[i] * dinv % r for i in range(cs.n_instance, nv)],
        "gamma_abc": [comb[i] * ginv % r for i in range(cs.n_instance)],
        "u": u, "v": v, "w": w, "zt": zt, "n": n,
    }


def groth16_prove_exponents(c: CurveParams, cs: R1CS, td: Groth16Trapdoor, ex, h: Sequence[int], r_: int, s_: int):
    """Discrete logs of the proof (A in G1, B in G2, C in G1) per create_proof_with_assignment."""
    r = c.fr.p
    z = cs.assignment()
    nv = len(z)
    A = (td.alpha + sum(z[i] * ex["u"][i] for i in range(nv)) + r_ * td.delta) % r
    B = (td.beta + sum(z[i] * ex["v"][i] for i in range(nv)) + s_ * td.delta) % r
    l_acc = sum(z[cs.n_instance + i] * ex["l_query"][i] for i in range(cs.n_witness)) % r
    h_acc = sum(hj * q for hj, q in zip(h, ex["h_query"])) % r
    C = (s_ * A + r_ * B - r_ * s_ % r * td.delta + l_acc + h_acc) % r
    return A, B, C


def groth16_check_exponents(c: CurveParams, cs: R1CS, td: Groth16Trapdoor, ex, A: int, B: int, C: int) -> bool:
    """the Groth16 pairing equation e(A,B) = e(alpha,beta) e(sum pub, gamma) e(C, delta), in the exponent"""
    r = c.fr.p
    z = cs.assignment()
    pub = sum(z[i] * ex["gamma_abc"][i] for i in range(cs.n_instance)) % r
    return (A * B - td.alpha * td.beta - pub * td.gamma - C * td.delta) % r == 0


# --------------------------------------------------------------------------------------------
# Pairing (row f4: Groth16::verify, plugins/arkworks/src/groth16.rs:459-466 -> ark_groth16::verify_proof ->
# E::miller_loop + final_exponentiation; the plugin's own pairing helpers and bilinearity tests are
# plugins/arkworks/src/pairing.rs:47-90,116-129).  Definition-level model: Fq12 = Fq[w]/(w^12 - a w^6 - b) as plain
# polynomials, G2 points mapped into E(Fq12) by the untwist, generic affine line functions, and the final
# exponentiation as one big power (p^12 - 1)/r.  Slow and simple on purpose: it is the uniqueness anchor for the
# tower-free C++ verifier.  Any non-degenerate bilinear pairing decides Groth16 verification identically.
# --------------------------------------------------------------------------------------------

PAIRING = {
    # w^12 = 2 w^6 - 2 (i = w^6 - 1, i^2 = -1); ate loop |x| = 0xd201000000010000
    "bls12_381": {"mod6": 2, "mod0": -2, "loop": 15132376222941642752, "i_shift": 1, "twist_div": True, "bn_tail": False},
    # w^12 = 18 w^6 - 82 (i = w^6 - 9); ate loop 6x + 2 with x = 4965661367192848881
    "bn254": {"mod6": 18, "mod0": -82, "loop": 29793968203157093288, "i_shift": 9, "twist_div": False, "bn_tail": True},
}


class Fq12Ctx:
    def __init__(self, c: CurveParams):
        self.c = c
        self.p = c.fq.p
        self.cfg = PAIRING[c.name]

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for k in range(22, 11, -1):  # w^k = w^(k-12) * (mod6 w^6 + mod0)
            v = t[k]
            if v:
                t[k - 6] += v * self.cfg["mod6"]
                t[k - 12] += v * self.cfg["mod0"]
        return [v % p for v in t[:12]]

    def one(self):
        return [1] + [0] * 11

    def scalar(self, v):
        return [v % self.p] + [0] * 11

    def add(self, a, b):
        return [(x + y) % self.p for x, y in zip(a, b)]

    def sub(self, a, b):
        return [(x - y) % self.p for x, y in zip(a, b)]

    def pow(self, a, e):
        r = self.one()
        base = a
        while e:
            if e & 1:
                r = self.mul(r, base)
            base = self.mul(base, base)
            e >>= 1
        return r

    def inv(self, a):
        """polynomial extended Euclid modulo the degree-12 modulus"""
        p = self.p
        mod = [(-self.cfg["mod0"]) % p, 0, 0, 0, 0, 0, (-self.cfg["mod6"]) % p, 0, 0, 0, 0, 0, 1]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(a) + [0], mod

        def deg(q):
            d = len(q) - 1
            while d and q[d] == 0:
                d -= 1
            return d

        def poly_div(aa, bb):
            da, db = deg(aa), deg(bb)
            temp = list(aa)
            o = [0] * len(aa)
            binv = pow(bb[db], -1, p)
            for i in range(da - db, -1, -1):
                o[i] = (o[i] + temp[db + i] * binv) % p
                for cidx in range(db + 1):
                    temp[cidx + i] = (temp[cidx + i] - o[i] * bb[cidx]) % p  # noqa
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % p
                    new[i + j] = (new[i + j] - low[i] * r[j]) % p
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, p)
        return [v * inv0 % p for v in lm[:12]]


def _embed_fq2(ctx: Fq12Ctx, a: F2):
    """a0 + a1 i with i = w^6 - i_shift"""
    s = ctx.cfg["i_shift"]
    out = [0] * 12
    out[0] = (a[0] - s * a[1]) % ctx.p
    out[6] = a[1] % ctx.p
    return out


def pairing_untwist(c: CurveParams, Q: Point2):
    """G2 (twist, Fq2 coordinates) -> E(Fq12)"""
    ctx = Fq12Ctx(c)
    x, y = _embed_fq2(ctx, Q[0]), _embed_fq2(ctx, Q[1])
    w = [0, 1] + [0] * 10
    w2, w3 = ctx.mul(w, w), ctx.mul(ctx.mul(w, w), w)
    if ctx.cfg["twist_div"]:
        return ctx.mul(x, ctx.inv(w2)), ctx.mul(y, ctx.inv(w3))
    return ctx.mul(x, w2), ctx.mul(y, w3)


def _e12_double(ctx, P):
    x, y = P
    m = ctx.mul(ctx.mul(ctx.scalar(3), ctx.mul(x, x)), ctx.inv(ctx.mul(ctx.scalar(2), y)))
    nx = ctx.sub(ctx.mul(m, m), ctx.mul(ctx.scalar(2), x))
    ny = ctx.sub(ctx.mul(m, ctx.sub(x, nx)), y)
    return nx, ny


def _e12_add(ctx, P, Q):
    if P[0] == Q[0] and P[1] == Q[1]:
        return _e12_double(ctx, P)
    m = ctx.mul(ctx.sub(Q[1], P[1]), ctx.inv(ctx.sub(Q[0], P[0])))
    nx = ctx.sub(ctx.sub(ctx.mul(m, m), P[0]), Q[0])
    ny = ctx.sub(ctx.mul(m, ctx.sub(P[0], nx)), P[1])
    return nx, ny


def _linefunc(ctx, P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = ctx.mul(ctx.sub(y2, y1), ctx.inv(ctx.sub(x2, x1)))
        return ctx.sub(ctx.mul(m, ctx.sub(xt, x1)), ctx.sub(yt, y1))
    if y1 == y2:
        m = ctx.mul(ctx.mul(ctx.scalar(3), ctx.mul(x1, x1)), ctx.inv(ctx.mul(ctx.scalar(2), y1)))
        return ctx.sub(ctx.mul(m, ctx.sub(xt, x1)), ctx.sub(yt, y1))
    return ctx.sub(xt, x1)


def pairing(c: CurveParams, P: Point, Q: Point2) -> List[int]:
    """e(P, Q) in Fq12 (12 coefficients); P in G1, Q in G2; the 1 for an infinity argument."""
    ctx = Fq12Ctx(c)
    if P is None or Q is None:
        return ctx.one()
    Q12 = pairing_untwist(c, Q)
    P12 = (ctx.scalar(P[0]), ctx.scalar(P[1]))
    R = Q12
    f = ctx.one()
    loop = ctx.cfg["loop"]
    for i in range(loop.bit_length() - 2, -1, -1):
        f = ctx.mul(ctx.mul(f, f), _linefunc(ctx, R, R, P12))
        R = _e12_double(ctx, R)
        if (loop >> i) & 1:
            f = ctx.mul(f, _linefunc(ctx, R, Q12, P12))
            R = _e12_add(ctx, R, Q12)
    if ctx.cfg["bn_tail"]:
        p = ctx.p
        Q1 = (ctx.pow(Q12[0], p), ctx.pow(Q12[1], p))
        nQ2 = (ctx.pow(Q1[0], p), ctx.sub([0] * 12, ctx.pow(Q1[1], p)))
        f = ctx.mul(f, _linefunc(ctx, R, Q1, P12))
        R = _e12_add(ctx, R, Q1)
        f = ctx.mul(f, _linefunc(ctx, R, nQ2, P12))
    return ctx.pow(f, (ctx.p ** 12 - 1) // c.fr.p)


def groth16_verify_pairing(c: CurveParams, vk: dict, public_inputs: Sequence[int], A: Point, B: Point2, C_: Point) -> bool:
    """ark_groth16::verify_proof: e(A, B) == e(alpha, beta) * e(sum_i x_i gamma_abc_i, gamma) * e(C, delta).
    vk: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (list, index 0 pairs with the constant ONE)."""
    ctx = Fq12Ctx(c)
    acc = vk["gamma_abc_g1"][0]
    for x, pt in zip(public_inputs, vk["gamma_abc_g1"][1:]):
        acc = g1_add(c, acc, g1_mul(c, x, pt))
    lhs = pairing(c, A, B)
    rhs = ctx.mul(ctx.mul(pairing(c, vk["alpha_g1"], vk["beta_g2"]), pairing(c, acc, vk["gamma_g2"])), pairing(c, C_, vk["delta_g2"]))
    return lhs == rhs


# ---- wire format: arkworks 0.3 CanonicalSerialize, compressed -------------------------------------------------
# Restated from the published format of ark-serialize / ark-ec / ark-ff 0.3 (reached from /root/reference/plugins/arkworks/src/
# groth16.rs:68-107): x as a little-endian canonical integer in ceil((MODULUS_BITS + 2) / 8) bytes, flags in the top two bits of the
# last byte (bit 7: y > -y, bit 6: infinity), Fq2 = c0 then c1 with the flags on c1 and the order "c1 first, then c0".
# UNPINNED: the reference holds no serialized vector; this is the checker for csrc/zl_serialize.h only.
def _fq_bytes(c: CurveParams) -> int:
    return (c.fq.bits + 2 + 7) // 8


def g1_compress(c: CurveParams, P: Point) -> bytes:
    nb = _fq_bytes(c)
    if P is None:
        return bytes(nb - 1) + bytes([0x40])
    x, y = P
    out = bytearray(x.to_bytes(nb, "little"))
    if y > (c.fq.p - y) % c.fq.p:
        out[-1] |= 0x80
    return bytes(out)


def fq_sqrt(p: int, a: int):
    """Tonelli-free: p = 3 mod 4 for both curves"""
    r = pow(a, (p + 1) // 4, p)
    return r if r * r % p == a % p else None


def g1_decompress(c: CurveParams, data: bytes) -> Point:
    nb, p = _fq_bytes(c), c.fq.p
    assert len(data) == nb
    flags = data[-1] & 0xC0
    x = int.from_bytes(data[:-1] + bytes([data[-1] & 0x3F]), "little")
    if flags & 0x40:
        return None
    y = fq_sqrt(p, (x * x * x + c.b) % p)
    assert y is not None
    if (y > p - y) != bool(flags & 0x80):
        y = p - y
    return (x, y)


def f2_gt(a: F2, b: F2) -> bool:  # ark-ff QuadExtField::cmp
    return (a[1], a[0]) > (b[1], b[0])


def g2_compress(c: CurveParams, P: Point2) -> bytes:
    nb, p = _fq_bytes(c), c.fq.p
    if P is None:
        return bytes(2 * nb - 1) + bytes([0x40])
    x, y = P
    out = bytearray(x[0].to_bytes(nb, "little") + x[1].to_bytes(nb, "little"))
    if f2_gt(y, ((p - y[0]) % p, (p - y[1]) % p)):
        out[-1] |= 0x80
    return bytes(out)


def f2_sqrt(p: int, a: F2):
    """Adj & Rodriguez-Henriquez, Alg. 9 (q = 3 mod 4, u^2 = -1) -- a different route than the backend's norm method"""
    def f2_pow(b, e):
        acc = (1, 0)
        while e:
            if e & 1:
                acc = f2_mul(p, acc, b)
            b = f2_mul(p, b, b)
            e >>= 1
        return acc
    if a == (0, 0):
        return (0, 0)
    a1 = f2_pow(a, (p - 3) // 4)
    alpha = f2_mul(p, a1, f2_mul(p, a1, a))
    a0 = f2_mul(p, (alpha[0], (p - alpha[1]) % p), alpha)  # alpha^q * alpha (conjugate = Frobenius)
    if a0 == (p - 1, 0):
        return None
    x0 = f2_mul(p, a1, a)
    if alpha == (p - 1, 0):
        r = ((p - x0[1]) % p, x0[0])  # u * x0
    else:
        b = f2_pow(((1 + alpha[0]) % p, alpha[1]), (p - 1) // 2)
        r = f2_mul(p, b, x0)
    return r if f2_mul(p, r, r) == a else None


def g2_decompress(c: CurveParams, data: bytes) -> Point2:
    nb, p = _fq_bytes(c), c.fq.p
    assert len(data) == 2 * nb
    flags = data[-1] & 0xC0
    x = (int.from_bytes(data[:nb], "little"), int.from_bytes(data[nb:-1] + bytes([data[-1] & 0x3F]), "little"))
    if flags & 0x40:
        return None
    rhs = f2_add(p, f2_mul(p, f2_mul(p, x, x), x), c.b2)
    y = f2_sqrt(p, rhs)
    assert y is not None
    ny = ((p - y[0]) % p, (p - y[1]) % p)
    if f2_gt(y, ny) != bool(flags & 0x80):
        y = ny
    return (x, y)


def groth16_proof_bytes(c: CurveParams, A: Point, B: Point2, C_: Point) -> bytes:
    return g1_compress(c, A) + g2_compress(c, B) + g1_compress(c, C_)


# ---- wire format: uncompressed points, ProvingKey / VerifyingKey (what the reference's ProvingContext codec writes, groth16.rs:142-179) ----
# Restated from the published ark-serialize / ark-ec / ark-groth16 0.3 layout: serialize_uncompressed (= serialize_unchecked) of an affine
# point is x then y with SWFlags on the last byte of y; a finite point carries SWFlags::default() = no bits, infinity is
# GroupAffine::zero() = (0, 1) with bit 6; a Vec is its u64 little-endian length followed by the elements; derived struct impls write the
# fields in declaration order.  UNPINNED like the compressed form above.
def g1_uncompressed(c: CurveParams, P: Point) -> bytes:
    nb = _fq_bytes(c)
    if P is None:
        out = bytearray(bytes(nb) + (1).to_bytes(nb, "little"))
        out[-1] |= 0x40
        return bytes(out)
    return P[0].to_bytes(nb, "little") + P[1].to_bytes(nb, "little")


def g2_uncompressed(c: CurveParams, P: Point2) -> bytes:
    nb = _fq_bytes(c)
    if P is None:
        out = bytearray(bytes(2 * nb) + (1).to_bytes(nb, "little") + bytes(nb))
        out[-1] |= 0x40
        return bytes(out)
    (x0, x1), (y0, y1) = P
    return b"".join(v.to_bytes(nb, "little") for v in (x0, x1, y0, y1))


def _vec(items: Sequence[bytes]) -> bytes:
    return len(items).to_bytes(8, "little") + b"".join(items)


def groth16_vk_bytes(c: CurveParams, vk: dict, compressed: bool = True) -> bytes:
    """ark_groth16::VerifyingKey: alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1"""
    e1 = g1_compress if compressed else g1_uncompressed
    e2 = g2_compress if compressed else g2_uncompressed
    return (e1(c, vk["alpha_g1"]) + e2(c, vk["beta_g2"]) + e2(c, vk["gamma_g2"]) + e2(c, vk["delta_g2"])
            + _vec([e1(c, P) for P in vk["gamma_abc_g1"]]))


def groth16_pk_bytes(c: CurveParams, pk: dict) -> bytes:
    """ark_groth16::ProvingKey, serialize_unchecked: vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query"""
    return (groth16_vk_bytes(c, pk["vk"], compressed=False) + g1_uncompressed(c, pk["beta_g1"]) + g1_uncompressed(c, pk["delta_g1"])
            + _vec([g1_uncompressed(c, P) for P in pk["a_query"]]) + _vec([g1_uncompressed(c, P) for P in pk["b_g1_query"]])
            + _vec([g2_uncompressed(c, P) for P in pk["b_g2_query"]]) + _vec([g1_uncompressed(c, P) for P in pk["h_query"]])
            + _vec([g1_uncompressed(c, P) for P in pk["l_query"]]))
