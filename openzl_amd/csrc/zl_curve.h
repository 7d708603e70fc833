// zl_curve.h -- short-Weierstrass (a = 0) group arithmetic in XYZZ coordinates, generic over the coordinate
// field (Fp<Fq> for G1, Fp2<Fq> for G2).  Host + device.
//
// Replaces, for this backend, ark-ec 0.3.0 short_weierstrass_jacobian::{add_assign_mixed, add_assign,
// double_in_place, into_affine} (surfaced by `pub use ec;`, /root/reference/plugins/arkworks/src/lib.rs:28-29;
// SURVEY.md §2.1).  arkworks uses Jacobian (X,Y,Z); this backend uses XYZZ (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2)
// because the mixed addition is 8M+2S instead of 7M+4S with one temporary fewer in registers.  Only the final
// *affine* point is ever compared with the reference path (unique, SURVEY.md §8 note N1).
// Formulas: EFD "xyzz" madd-2008-s / add-2008-s / dbl-2008-s-1 / mdbl-2008-s-1.
#pragma once
#include "zl_field.h"
#include "zl_field28.h"

// ---- Fq2 = Fq[u]/(u^2+1) (both curves), the G2 coordinate field -----------------------------------------
template <class P>
struct Fp2 {
    Fp<P> c0, c1;
    ZL_HD static Fp2 zero() { return Fp2{Fp<P>::zero(), Fp<P>::zero()}; }
    ZL_HD static Fp2 one() { return Fp2{Fp<P>::one(), Fp<P>::zero()}; }
    ZL_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZL_HD bool raw_zero() const { return c0.raw_zero() && c1.raw_zero(); }
    ZL_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    ZL_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
};
namespace zl {
template <class P> ZL_HD Fp2<P> add(const Fp2<P>& a, const Fp2<P>& b) { return Fp2<P>{add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <class P> ZL_HD Fp2<P> sub(const Fp2<P>& a, const Fp2<P>& b) { return Fp2<P>{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
template <class P> ZL_HD Fp2<P> dbl(const Fp2<P>& a) { return Fp2<P>{dbl(a.c0), dbl(a.c1)}; }
template <class P> ZL_HD Fp2<P> neg(const Fp2<P>& a) { return Fp2<P>{neg(a.c0), neg(a.c1)}; }
template <int J, class P> ZL_HD Fp2<P> subk(const Fp2<P>& a, const Fp2<P>& b) { return sub(a, b); }
template <int J, class P> ZL_HD Fp2<P> negk(const Fp2<P>& a) { return neg(a); }
template <class P> ZL_HD Fp2<P> wred(const Fp2<P>& a) { return a; }
template <class P> ZL_HD Fp<P> muladd(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) { return add(mul(a, b), mul(c, d)); }
template <class P> ZL_HD Fp2<P> canon(const Fp2<P>& a) { return a; }
template <class P> ZL_HD Fp2<P> mul(const Fp2<P>& a, const Fp2<P>& b) {
    // Karatsuba: 3 base multiplications
    Fp<P> v0 = mul(a.c0, b.c0), v1 = mul(a.c1, b.c1);
    Fp<P> s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fp2<P>{sub(v0, v1), sub(sub(s, v0), v1)};
}
template <class P> ZL_HD Fp2<P> muladd(const Fp2<P>& a, const Fp2<P>& b, const Fp2<P>& c, const Fp2<P>& d) { return add(mul(a, b), mul(c, d)); }
template <class P> ZL_HD Fp2<P> sqr(const Fp2<P>& a) {
    // (c0+c1)(c0-c1) + 2 c0 c1 u
    Fp<P> t = mul(add(a.c0, a.c1), sub(a.c0, a.c1));
    Fp<P> m = mul(a.c0, a.c1);
    return Fp2<P>{t, dbl(m)};
}
template <class P> ZL_HD Fp2<P> inv(const Fp2<P>& a) {
    Fp<P> n = inv(add(sqr(a.c0), sqr(a.c1)));
    return Fp2<P>{mul(a.c0, n), neg(mul(a.c1, n))};
}
template <class P> ZL_HD Fp2<P> to_mont(const Fp2<P>& a) { return Fp2<P>{to_mont(a.c0), to_mont(a.c1)}; }
template <class P> ZL_HD Fp2<P> from_mont(const Fp2<P>& a) { return Fp2<P>{from_mont(a.c0), from_mont(a.c1)}; }
}  // namespace zl

template <class P>
struct FieldIO<Fp2<P>> {
    static constexpr int WORDS = 2 * P::N;
    ZL_HD static Fp2<P> load_canon(const uint32_t* w) { return Fp2<P>{FieldIO<Fp<P>>::load_canon(w), FieldIO<Fp<P>>::load_canon(w + P::N)}; }
    ZL_HD static Fp2<P> load_mont32(const uint32_t* w) { return Fp2<P>{FieldIO<Fp<P>>::load_mont32(w), FieldIO<Fp<P>>::load_mont32(w + P::N)}; }
    ZL_HD static void store_canon(uint32_t* w, const Fp2<P>& a) { FieldIO<Fp<P>>::store_canon(w, a.c0); FieldIO<Fp<P>>::store_canon(w + P::N, a.c1); }
};

// ---- Fq2 over the lazily reduced 28-bit base field (BLS12-381 G2) ------------------------------------------------------------------
// Same contracts as Fp28 component-wise (zl_field28.h): mul / sqr return components < 2q, add / subk<J> track bounds, so the point
// formulas below need no second set of annotations.  Two flavours with one memory layout: INL = false calls the out-of-line Fq2
// product routines (small code: every kernel but one), INL = true inlines the product scans -- used by the bucket accumulation
// only, where it is worth 2 ms per 2^20 MSM; inlining it everywhere costs 6 minutes of compile time for no measurable gain.
template <class B, bool INL>
struct Fp2LT {
    B c0, c1;
    ZL_HD static constexpr Fp2LT zero() { return Fp2LT{B::zero(), B::zero()}; }
    ZL_HD static constexpr Fp2LT one() { return Fp2LT{B::one(), B::zero()}; }
    ZL_HD constexpr bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZL_HD constexpr bool raw_zero() const { return c0.raw_zero() && c1.raw_zero(); }
    ZL_HD bool operator==(const Fp2LT& o) const { return c0 == o.c0 && c1 == o.c1; }
    ZL_HD bool operator!=(const Fp2LT& o) const { return !(*this == o); }
};
template <class B> using Fp2L = Fp2LT<B, false>;
// the flavour a hot kernel computes in (identity for every other field)
template <class F> struct HotField { using type = F; };
template <class B> struct HotField<Fp2LT<B, false>> { using type = Fp2LT<B, true>; };
namespace zl {
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> add(const Fp2LT<B, I>& a, const Fp2LT<B, I>& b) { return Fp2LT<B, I>{add(a.c0, b.c0), add(a.c1, b.c1)}; }
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> dbl(const Fp2LT<B, I>& a) { return Fp2LT<B, I>{dbl(a.c0), dbl(a.c1)}; }
template <int J, class B, bool I> ZL_HD constexpr Fp2LT<B, I> subk(const Fp2LT<B, I>& a, const Fp2LT<B, I>& b) { return Fp2LT<B, I>{subk<J>(a.c0, b.c0), subk<J>(a.c1, b.c1)}; }
template <int J, class B, bool I> ZL_HD constexpr Fp2LT<B, I> negk(const Fp2LT<B, I>& a) { return Fp2LT<B, I>{negk<J>(a.c0), negk<J>(a.c1)}; }
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> sub(const Fp2LT<B, I>& a, const Fp2LT<B, I>& b) { return subk<4>(a, b); }
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> neg(const Fp2LT<B, I>& a) { return negk<4>(a); }
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> wred(const Fp2LT<B, I>& a) { return Fp2LT<B, I>{wred(a.c0), wred(a.c1)}; }
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> canon(const Fp2LT<B, I>& a) { return Fp2LT<B, I>{canon(a.c0), canon(a.c1)}; }
// ---- operands that ONLY feed a product scan ---------------------------------------------------------------------------------------
// The kernels are bound by instruction issue and the carry pass of a lazy add / sub is 39 ordinary instructions, so an operand that goes
// straight into a product scan (never a squaring, a zero test, another subtraction or memory) may skip it: zl_field28.h overloads these
// for the 28-bit field on the device ("un-carried": the scans take any 32-bit limbs as long as their 64-bit columns hold).  The price is
// one bias step more (2^J q with the subtrahend < 2^(J-1) q), which the product budget absorbs.  Everywhere else -- host code, the 32-bit
// field, Fq2 (its products negate an operand component internally, which needs carried limbs <= 16q) -- these are the carried forms.
template <class F> struct ScanBias { static constexpr int J = 3; };  // bias step of the mixed addition's un-stored differences (8q: the carried forms)
template <class A, class B> struct ScanBias<Fp28<A, B>> { static constexpr int J = 4; };
template <int J, class F> ZL_HD constexpr F subk_scan(const F& a, const F& b) { return subk<J>(a, b); }
template <int J, class F> ZL_HD constexpr F negk_scan(const F& b) { return negk<J>(b); }
// r^2 - ppp - 2q + 6q (the x coordinate of a sum): < 8 for r^2, ppp, q < 2
template <class F> ZL_HD constexpr F x3_of(const F& rr, const F& ppp, const F& q) { return subk<2>(subk<1>(rr, ppp), dbl(q)); }

// out-of-line Fq2 products (the INL = false flavour): one call per product, operands as scalar words (zl_field28.h)
template <class A, class P> ZL_HD Fp2LT<Fp28<A, P>, false> fq2_mul_called(const Fp2LT<Fp28<A, P>, false>& a, const Fp2LT<Fp28<A, P>, false>& b) {
    Fp2LT<Fp28<A, P>, false> r;
    if constexpr (A::L == 14) unpair28(fq2_mul_call28<A, P>(ZL_A14(a.c0), ZL_A14(a.c1), ZL_A14(b.c0), ZL_A14(b.c1)), r.c0, r.c1);
    else unpair28(fq2_mul_call28x10<A, P>(ZL_A10(a.c0), ZL_A10(a.c1), ZL_A10(b.c0), ZL_A10(b.c1)), r.c0, r.c1);
    return r;
}
template <class A, class P> ZL_HD Fp2LT<Fp28<A, P>, false> fq2_sqr_called(const Fp2LT<Fp28<A, P>, false>& a) {
    Fp2LT<Fp28<A, P>, false> r;
    if constexpr (A::L == 14) unpair28(fq2_sqr_call28<A, P>(ZL_A14(a.c0), ZL_A14(a.c1)), r.c0, r.c1);
    else unpair28(fq2_sqr_call28x10<A, P>(ZL_A10(a.c0), ZL_A10(a.c1)), r.c0, r.c1);
    return r;
}
// (a0 + a1 u)(b0 + b1 u): each component ONE dual product scan; operand components <= 16q, result components < 2q; -b1 is a scan-only
// operand (32q - b1, un-carried on the device)
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> mul(const Fp2LT<B, I>& a, const Fp2LT<B, I>& b) {
    if constexpr (I) return Fp2LT<B, I>{muladd(a.c0, b.c0, a.c1, negk_scan<5>(b.c1)), muladd(a.c0, b.c1, a.c1, b.c0)};  // 16*16 + 16*32 <= 2500
    else return fq2_mul_called(a, b);
}
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> sqr(const Fp2LT<B, I>& a) {
    if constexpr (I) return Fp2LT<B, I>{mul(add(a.c0, a.c1), subk<4>(a.c0, a.c1)), mul(dbl(a.c0), a.c1)};
    else return fq2_sqr_called(a);
}
// a b + c d.  Inlined flavour: each component is ONE four-product scan (980 mads) -> components < 2q; operands' components <= 16q.
// Called flavour: two calls and a lazy add (components < 4q; a four-product call would need 112 argument words).
template <class B, bool I>
ZL_HD constexpr Fp2LT<B, I> muladd(const Fp2LT<B, I>& a, const Fp2LT<B, I>& b, const Fp2LT<B, I>& c, const Fp2LT<B, I>& d) {
    if constexpr (I) {
        return Fp2LT<B, I>{muladd4(a.c0, b.c0, a.c1, negk_scan<5>(b.c1), c.c0, d.c0, c.c1, negk_scan<5>(d.c1)),  // 2 * (16*16 + 16*32) <= 2500
                           muladd4(a.c0, b.c1, a.c1, b.c0, c.c0, d.c1, c.c1, d.c0)};
    } else {
        return add(mul(a, b), mul(c, d));
    }
}
template <class B, bool I> ZL_HD constexpr Fp2LT<B, I> inv(const Fp2LT<B, I>& a) {
    const B n = inv(add(sqr(a.c0), sqr(a.c1)));
    return Fp2LT<B, I>{mul(a.c0, n), mul(negk<4>(a.c1), n)};
}
}  // namespace zl
template <class B, bool I>
struct FieldIO<Fp2LT<B, I>> {
    static constexpr int WORDS = 2 * FieldIO<B>::WORDS;
    ZL_HD static Fp2LT<B, I> load_canon(const uint32_t* w) { return Fp2LT<B, I>{FieldIO<B>::load_canon(w), FieldIO<B>::load_canon(w + FieldIO<B>::WORDS)}; }
    ZL_HD static Fp2LT<B, I> load_mont32(const uint32_t* w) { return Fp2LT<B, I>{FieldIO<B>::load_mont32(w), FieldIO<B>::load_mont32(w + FieldIO<B>::WORDS)}; }
    ZL_HD static void store_canon(uint32_t* w, const Fp2LT<B, I>& a) { FieldIO<B>::store_canon(w, a.c0); FieldIO<B>::store_canon(w + FieldIO<B>::WORDS, a.c1); }
};

// ---- points ----------------------------------------------------------------------------------------------
// Affine point in device memory.  Infinity is encoded as x = y = 0 ((0,0) is on neither curve: b != 0).
template <class F>
struct Affine {
    F x, y;
    ZL_HD constexpr bool is_inf() const { return x.raw_zero() && y.raw_zero(); }  // stored points are canonical: all-zero limbs
    ZL_HD static constexpr Affine inf() { return Affine{F::zero(), F::zero()}; }
};
template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    // exact-zero limbs: infinity is only ever written as zero(); a computed zz is a product of non-zero factors (P == 0 and
    // y == 0 are branched away before multiplying; both curves have odd order, so no point has y == 0)
    ZL_HD constexpr bool is_inf() const { return zz.raw_zero(); }
    ZL_HD static constexpr XYZZ inf() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }
    ZL_HD static constexpr XYZZ from_affine(const Affine<F>& a) {
        if (a.is_inf()) return inf();
        return XYZZ{a.x, a.y, F::one(), F::one()};
    }
};

namespace zl {

// Bounds in the comments are in units of q for the lazily reduced field (zl_field28.h); for the fully reduced 32-bit field the
// subk<J> / wred spellings are plain sub / identity.  Contract of every routine: coordinates < 8q in, < 8q out.
// 2*(x,y) for an affine, non-infinity point (mdbl-2008-s-1, a = 0)
template <class F>
ZL_HD constexpr XYZZ<F> dbl_affine(const F& x, const F& y) {          // x, y < 8
    const F u = dbl(y);                                       // < 16
    const F v = sqr(u), w = mul(u, v), s = mul(x, v);         // 256, 32, 16 -> < 2
    const F xx = sqr(x);                                      // 64 -> < 2
    const F m = add(dbl(xx), xx);                             // < 6
    const F x3 = subk<2>(sqr(m), dbl(s));                     // 36 -> 2; dbl(s) < 4 -> < 6
    const F y3 = muladd(m, subk<3>(s, x3), w, negk<3>(y));    // m (s - x3) - w y, one reduction: 6*10 + 2*8 -> < 2
    return XYZZ<F>{x3, y3, v, w};  // y == 0 -> zz == 0 -> infinity (does not occur on these curves)
}
// p = 2p (dbl-2008-s-1, a = 0)
template <class F>
ZL_HD constexpr void dbl_inplace(XYZZ<F>& p) {
    if (p.is_inf()) return;
    const F u = dbl(p.y);                                     // < 16
    const F v = sqr(u), w = mul(u, v), s = mul(p.x, v);       // < 2
    const F xx = sqr(p.x);
    const F m = add(dbl(xx), xx);                             // < 6
    const F x3 = subk<2>(sqr(m), dbl(s));                     // < 6
    p.y = muladd(m, subk<3>(s, x3), w, negk<3>(p.y));         // 60 + 16 -> < 2
    p.x = x3;
    p.zz = mul(v, p.zz);                                      // 16 -> < 2
    p.zzz = mul(w, p.zzz);
}
// p += (qx, qy) (affine, canonical or < 2q, q must not be infinity); neg_q selects p -= q.   madd-2008-s
template <class F>
ZL_HD constexpr void add_mixed(XYZZ<F>& p, const F& qx, const F& qy_in, bool neg_q) {
    constexpr int J = ScanBias<F>::J;                         // 3 (carried: bounds as annotated) or 4 (28-bit field: un-carried, in brackets)
    if (p.is_inf()) {
        p.x = qx; p.y = neg_q ? negk<1>(qy_in) : qy_in; p.zz = F::one(); p.zzz = F::one();
        return;
    }
    const F qy = neg_q ? negk_scan<2>(qy_in) : qy_in;         // < 4: feeds the product with zzz only
    const F u2 = mul(qx, p.zz), s2 = mul(qy, p.zzz);          // 16, 32 -> < 2
    const F pp_ = subk<3>(u2, p.x), r = subk<3>(s2, p.y);     // < 10 (squared / tested below: carried)
    if (pp_.is_zero()) {
        if (r.is_zero()) { p = dbl_affine(qx, neg_q ? negk<1>(qy_in) : qy_in); return; }
        p = XYZZ<F>::inf();
        return;
    }
    const F pp = sqr(pp_), ppp = mul(pp_, pp), q = mul(p.x, pp);  // 100, 20, 16 -> < 2
    const F x3 = x3_of(sqr(r), ppp, q);                       // (2 + 2) + 4 -> < 8 (stored: carried)
    p.y = muladd(r, subk_scan<J>(q, x3), negk_scan<J>(p.y), ppp);  // r (q - x3) - y1 ppp, one reduction: 10*10 + 8*2 [10*18 + 16*2] -> < 2
    p.x = x3;
    p.zz = mul(p.zz, pp);                                     // < 2
    p.zzz = mul(p.zzz, ppp);
}
// p += q (add-2008-s)
template <class F>
ZL_HD constexpr void add_full(XYZZ<F>& p, const XYZZ<F>& q) {
    if (q.is_inf()) return;
    if (p.is_inf()) { p = q; return; }
    const F u1 = mul(p.x, q.zz), u2 = mul(q.x, p.zz);         // 64 -> < 2
    const F s1 = mul(p.y, q.zzz), s2 = mul(q.y, p.zzz);
    const F pp_ = subk<1>(u2, u1), r = subk<1>(s2, s1);       // < 4
    if (pp_.is_zero()) {
        if (r.is_zero()) { dbl_inplace(p); return; }
        p = XYZZ<F>::inf();
        return;
    }
    const F pp = sqr(pp_), ppp = mul(pp_, pp), q_ = mul(u1, pp);  // < 2
    constexpr int J = ScanBias<F>::J;
    const F x3 = x3_of(sqr(r), ppp, q_);                      // < 8
    p.y = muladd(r, subk_scan<J>(q_, x3), negk_scan<2>(s1), ppp);  // 4*10 + 4*2 [4*18 + 4*2] -> < 2
    p.x = x3;
    p.zz = mul(mul(p.zz, q.zz), pp);
    p.zzz = mul(mul(p.zzz, q.zzz), ppp);
}
// ---- Jacobian coordinates (x = X/Z^2, y = Y/Z^3) for long doubling chains: dbl-2009-l is 2M + 5S (5.85 multiplication-equivalents on the
// 28-bit field) against 7.8 for the XYZZ doubling.  Used by the table construction (c doublings per level).  Z == 0 encodes infinity.
template <class F>
struct Jac {
    F x, y, z;
    ZL_HD constexpr bool is_inf() const { return z.raw_zero(); }
    ZL_HD static constexpr Jac inf() { return Jac{F::one(), F::one(), F::zero()}; }
    ZL_HD static constexpr Jac from_affine(const Affine<F>& a) {
        if (a.is_inf()) return inf();
        return Jac{a.x, a.y, F::one()};
    }
};
// p = 2p (dbl-2009-l, a = 0).  Coordinates < 8q in, < 8q out (bounds in units of q beside every line; proved in zl_bounds.h)
template <class F>
ZL_HD constexpr void jac_dbl_inplace(Jac<F>& p) {
    if (p.is_inf()) return;
    const F a = sqr(p.x), b = sqr(p.y);                       // 64 -> < 2
    const F c = sqr(b);                                       // < 2
    const F t = add(p.x, b);                                  // < 10
    const F d = dbl(subk<1>(subk<1>(sqr(t), a), c));          // (2 + 2) + 2 = 6 -> < 12
    const F e = add(dbl(a), a);                               // < 6
    const F x3 = wred(subk<5>(sqr(e), dbl(d)));               // 2 + 32 = 34 -> weakly reduced < 4
    const F c8 = dbl(dbl(dbl(c)));                            // < 16
    const F z3 = mul(dbl(p.y), p.z);                          // 16 * 8 -> < 2
    p.y = muladd(e, subk<2>(d, x3), negk<4>(c8), F::one());   // e (d - x3) - 8c: 6 * 16 + 16 * 1 -> < 2 (Fq2 operands must stay <= 16)
    p.x = x3;
    p.z = z3;
}
// p = 2^n p.  A run of doublings is cheaper in Jacobian coordinates (dbl-2009-l: 2M + 5S against 6M + 3S + a dual scan for the XYZZ doubling);
// XYZZ -> Jacobian with Z = zzz is (X zz^2, Y zzz^2, zzz) (x = X / zz, zz^3 = zzz^2), back is (X, Y, Z^2, Z^3): six multiplications for the
// round trip, so runs of at least four doublings take it.  Used by the host Horner over the windows of an MSM (c doublings per window).
// Bounds: XYZZ coordinates <= 8q in -> products of (8, 2) -> Jacobian coordinates < 2q; jac_dbl_inplace keeps them < 8q; out x < 4q, y, zz, zzz < 2q.
template <class F>
ZL_HD constexpr void dbl_n(XYZZ<F>& p, int n) {
    if (p.is_inf() || n <= 0) return;
    if (n < 4) {
        for (int k = 0; k < n; k++) dbl_inplace(p);
        return;
    }
    Jac<F> j{mul(p.x, sqr(p.zz)), mul(p.y, sqr(p.zzz)), p.zzz};
    for (int k = 0; k < n; k++) jac_dbl_inplace(j);
    const F zz = sqr(j.z);
    p = XYZZ<F>{j.x, j.y, zz, mul(zz, j.z)};
}
// p = -p
template <class F>
ZL_HD constexpr void neg_inplace(XYZZ<F>& p) {
    p.y = wred(negk<3>(p.y));                                 // y < 8 -> 8q - y -> weakly reduced < 4
}
template <class F>
ZL_HD constexpr Affine<F> to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>::inf();
    const F izzz = inv(p.zzz);
    const F t = mul(p.zz, izzz);  // zz/zzz = 1/z
    const F izz = sqr(t);         // zz^2/zzz^2 = 1/zz  (zz^3 = zzz^2)
    return Affine<F>{canon(mul(p.x, izz)), canon(mul(p.y, izzz))};  // canonical: the all-zero limb test stays exact in memory
}
// p = k*p for a small unsigned multiplier (double-and-add, MSB first)
template <class F>
ZL_HD XYZZ<F> mul_small(const XYZZ<F>& p, uint32_t k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 31; i >= 0; i--) {
        dbl_inplace(acc);
        if ((k >> i) & 1) add_full(acc, p);
    }
    return acc;
}
// k*p for a 256-bit little-endian scalar in 8 words (host-side tails, generator kernels)
template <class F>
ZL_HD XYZZ<F> mul_scalar(const XYZZ<F>& p, const uint32_t* k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        dbl_inplace(acc);
        if ((k[i >> 5] >> (i & 31)) & 1) add_full(acc, p);
    }
    return acc;
}
// The same product with 4-bit windows for the host tails of a proof (a handful of 255-bit scalar multiplications per proof, each on the critical
// path of a small one): table 1..15 p (14 additions), then 64 x (four doublings through Jacobian coordinates, dbl_n, + one table addition):
// ~3000 multiplication-equivalents instead of ~4100 for double-and-add.
template <class F>
XYZZ<F> mul_scalar_w4(const XYZZ<F>& p, const uint32_t* k) {
    XYZZ<F> tab[16];
    tab[0] = XYZZ<F>::inf();
    tab[1] = p;
    for (int d = 2; d < 16; d++) {
        tab[d] = tab[d - 1];
        add_full(tab[d], p);
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 63; w >= 0; w--) {
        dbl_n(acc, 4);
        const uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) add_full(acc, tab[d]);
    }
    return acc;
}
// k1*p + k2*q for two 256-bit little-endian scalars (Shamir's trick: one doubling chain, additions from {p, q, p+q}); host tails
template <class F>
ZL_HD XYZZ<F> mul_scalar2(const XYZZ<F>& p, const uint32_t* k1, const XYZZ<F>& q, const uint32_t* k2) {
    XYZZ<F> pq = p;
    add_full(pq, q);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int i = 255; i >= 0; i--) {
        dbl_inplace(acc);
        const int b1 = (k1[i >> 5] >> (i & 31)) & 1, b2 = (k2[i >> 5] >> (i & 31)) & 1;
        if (b1 && b2) add_full(acc, pq);
        else if (b1) add_full(acc, p);
        else if (b2) add_full(acc, q);
    }
    return acc;
}
}  // namespace zl
template <class F> using Jac = zl::Jac<F>;  // global spelling, like Affine / XYZZ
#include "zl_bounds.h"  // compile-time proof of the lazy-reduction contracts of the formulas above
