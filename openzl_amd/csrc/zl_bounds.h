// zl_bounds.h -- compile-time proof that the point formulas of zl_curve.h respect the lazy-reduction contracts of zl_field28.h.
//
// The 14 x 28-bit field never reduces sums and differences; every routine has a contract in units of q (zl_field28.h:7-14).  Instead of
// trusting the bound comments beside the formulas, the formulas themselves (they are templates over the coordinate field) are instantiated
// here with an ABSTRACT field `zl::BF` that carries only "value <= b*q" and evaluated inside static_asserts: a contract violation calls a
// non-constexpr function, which makes the static_assert's condition a non-constant expression -> the translation unit does not compile.
// Covered: dbl_affine, dbl_inplace, jac_dbl_inplace, add_mixed (both signs, and its doubling branch), add_full (and its doubling branch), neg_inplace,
// to_affine -- over the base field (G1) and over Fq2 in both flavours (inlined four-product scans / called dual scans) (G2) -- with
// every coordinate at the contract's maximum (8q), plus the closure property: results are <= 8q again, so any sequence of group
// operations stays inside the contracts.  tests/test_field28_bounds.py drives the real arithmetic at the same bounds (host + device).
// The constants below are those of the 14-limb BLS12-381 instance, the tighter one; the 10-limb BN254 instance (round 4) has 26 spare bits instead of 11, and
// gen_params.py asserts for every instance it emits that 2500 q < R' and that the top limb of 40000 q fits 32 bits, so one proof covers both.
#pragma once

namespace zl {
void bound_contract_violated();  // never defined: reaching it in a constant expression is the compile error

struct BF {
    int b;  // value <= b * q
    static constexpr BF zero() { return BF{0}; }
    static constexpr BF one() { return BF{1}; }
    // the generic (non-exceptional) path of every formula is the one checked; the exceptional branches are checked on their own below
    constexpr bool raw_zero() const { return false; }
    constexpr bool is_zero() const {
        if (b > 2000) bound_contract_violated();  // is_zero / wred: value <= 2000 q
        return false;
    }
};
constexpr int BF_MUL_MAX = 2500;   // a*b < 2^392 q  <=  B(a) B(b) <= 2500   (2500 q < 2^392; also keeps every limb of an operand < 2^28)
constexpr int BF_TOP_MAX = 40000;  // limbs are u32: the top limb (weight 2^364) of b*q must stay below 2^32
constexpr BF bf_chk(int b) {
    if (b > BF_TOP_MAX) bound_contract_violated();
    return BF{b};
}
constexpr BF add(const BF& a, const BF& b) { return bf_chk(a.b + b.b); }
constexpr BF dbl(const BF& a) { return bf_chk(2 * a.b); }
template <int J>
constexpr BF subk(const BF& a, const BF& b) {
    static_assert(J >= 1 && J <= 6, "bias table holds 2q .. 64q");
    if (b.b > (1 << J)) bound_contract_violated();  // a - b + 2^J q must be non-negative
    return bf_chk(a.b + (1 << J));
}
template <int J>
constexpr BF negk(const BF& a) { return subk<J>(BF::zero(), a); }
constexpr BF sub(const BF& a, const BF& b) { return subk<4>(a, b); }
constexpr BF neg(const BF& a) { return negk<4>(a); }
constexpr BF mul(const BF& a, const BF& b) {
    if (a.b * b.b > BF_MUL_MAX) bound_contract_violated();
    return BF{2};
}
constexpr BF sqr(const BF& a) { return mul(a, a); }
constexpr BF muladd(const BF& a, const BF& b, const BF& c, const BF& d) {
    if (a.b * b.b + c.b * d.b > BF_MUL_MAX) bound_contract_violated();
    return BF{2};
}
constexpr BF muladd4(const BF& a, const BF& b, const BF& c, const BF& d, const BF& e, const BF& f, const BF& g, const BF& h) {
    if (a.b * b.b + c.b * d.b + e.b * f.b + g.b * h.b > BF_MUL_MAX) bound_contract_violated();
    return BF{2};
}
constexpr BF wred(const BF& a) {
    if (a.b > 2000) bound_contract_violated();
    return BF{4};
}
constexpr BF canon(const BF& a) { return BF{wred(a).b > 0 ? 1 : 0}; }
constexpr BF inv(const BF& a) {  // Fermat ladder: acc = sqr(acc); acc = mul(acc, a) with acc < 2q throughout
    return mul(sqr(BF{2}), a);
}
// scan-only operands of the 28-bit field (zl_field28.h subk_scan / negk_scan): un-carried on the device, so even the TOP limb of
// 2^J q - b must be non-negative without a carry pass: b <= 2^(J-1) q
template <int J>
constexpr BF subk_scan(const BF& a, const BF& b) {
    static_assert(J >= 2 && J <= 6, "bias table holds 2q .. 64q");
    if (2 * b.b > (1 << J)) bound_contract_violated();
    return bf_chk(a.b + (1 << J));
}
template <int J>
constexpr BF negk_scan(const BF& b) { return subk_scan<J>(BF::zero(), b); }
template <> struct ScanBias<BF> { static constexpr int J = 4; };  // the G1 model follows the 28-bit field
// the called Fq2 flavour computes the same dual scans out of line
constexpr Fp2LT<BF, false> fq2_mul_called(const Fp2LT<BF, false>& a, const Fp2LT<BF, false>& b) {
    return Fp2LT<BF, false>{muladd(a.c0, b.c0, a.c1, negk_scan<5>(b.c1)), muladd(a.c0, b.c1, a.c1, b.c0)};
}
constexpr Fp2LT<BF, false> fq2_sqr_called(const Fp2LT<BF, false>& a) {
    return Fp2LT<BF, false>{mul(add(a.c0, a.c1), subk<4>(a.c0, a.c1)), mul(dbl(a.c0), a.c1)};
}

namespace bounds {
constexpr int COORD_MAX = 8;  // contract of every point routine: coordinates <= 8q in, <= 8q out
constexpr int AFFINE_MAX = 2; // affine operands of add_mixed: canonical in memory, < 2q after an optional negation
constexpr int hi(const BF& a) { return a.b; }
template <class B, bool I> constexpr int hi(const Fp2LT<B, I>& a) { return a.c0.b > a.c1.b ? a.c0.b : a.c1.b; }
template <class F> struct Mk;
template <> struct Mk<BF> { static constexpr BF at(int b) { return BF{b}; } };
template <class B, bool I> struct Mk<Fp2LT<B, I>> { static constexpr Fp2LT<B, I> at(int b) { return Fp2LT<B, I>{BF{b}, BF{b}}; } };
template <class F> constexpr bool closed(const XYZZ<F>& p) { return hi(p.x) <= COORD_MAX && hi(p.y) <= COORD_MAX && hi(p.zz) <= COORD_MAX && hi(p.zzz) <= COORD_MAX; }
template <class F> constexpr XYZZ<F> worst() { return XYZZ<F>{Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX)}; }

template <class F>
constexpr bool formulas_hold() {
    bool ok = true;
    {   // mixed addition, both signs
        XYZZ<F> p = worst<F>();
        add_mixed(p, Mk<F>::at(AFFINE_MAX), Mk<F>::at(AFFINE_MAX), false);
        ok = ok && closed(p);
        XYZZ<F> m = worst<F>();
        add_mixed(m, Mk<F>::at(AFFINE_MAX), Mk<F>::at(AFFINE_MAX), true);
        ok = ok && closed(m);
    }
    {   // its P == Q branch, and the doubling of an affine point with coordinates at the routine's stated maximum
        ok = ok && closed(dbl_affine(Mk<F>::at(AFFINE_MAX), Mk<F>::at(AFFINE_MAX)));
        ok = ok && closed(dbl_affine(Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX)));
    }
    {   // full addition and its P == Q branch
        XYZZ<F> p = worst<F>();
        add_full(p, worst<F>());
        ok = ok && closed(p);
        XYZZ<F> d = worst<F>();
        dbl_inplace(d);
        ok = ok && closed(d);
    }
    {   // Jacobian doubling chain (table construction)
        Jac<F> j{Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX), Mk<F>::at(COORD_MAX)};
        jac_dbl_inplace(j);
        ok = ok && hi(j.x) <= COORD_MAX && hi(j.y) <= COORD_MAX && hi(j.z) <= COORD_MAX;
    }
    {   // runs of doublings through Jacobian coordinates (host Horner over the windows): both the short XYZZ form and the round trip
        XYZZ<F> s3 = worst<F>();
        dbl_n(s3, 3);
        ok = ok && closed(s3);
        XYZZ<F> s5 = worst<F>();
        dbl_n(s5, 5);
        ok = ok && closed(s5);
    }
    {   // negation, normalisation
        XYZZ<F> p = worst<F>();
        neg_inplace(p);
        ok = ok && closed(p);
        const Affine<F> a = to_affine(worst<F>());
        ok = ok && hi(a.x) <= 1 && hi(a.y) <= 1;
    }
    return ok;
}
static_assert(formulas_hold<BF>(), "G1 point formulas violate the lazy-reduction contracts of zl_field28.h");
static_assert(formulas_hold<Fp2LT<BF, true>>(), "G2 point formulas (inlined Fq2 scans) violate the lazy-reduction contracts");
static_assert(formulas_hold<Fp2LT<BF, false>>(), "G2 point formulas (called Fq2 scans) violate the lazy-reduction contracts");
}  // namespace bounds
}  // namespace zl
