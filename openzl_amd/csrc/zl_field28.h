// zl_field28.h -- base fields on unsaturated 28-bit limbs with lazy reduction (gfx950 MSM kernels): BLS12-381 Fq on 14 limbs (R' = 2^392), and from
// round 4 BN254 Fq on 10 limbs (R' = 2^280: the same contracts hold a fortiori -- q < 2^254 leaves 26 spare bits, gen_params.py asserts the limits the
// bound proofs of zl_bounds.h assume).  The text below speaks of the 14-limb instance.
//
// Why: v_mad_u64_u32 accumulates 64 bits; with 28-bit limbs a column of <= 28 products (< 2^56 each) never overflows, so the
// Montgomery product is a pure chain of 392 mads with no carry handling at all (the 32-bit-limb multiplier needs a
// v_addc_co_u32 after every mad).  Measured (tools/fbench28.hip): 66.8 vs 49.6 G mul/s at 2 waves/SIMD, 71.6 vs 55.3 at 4.
// Representation: value * 2^392 mod q (Montgomery, R' = 2^392), limbs < 2^28, value only *weakly* reduced: q < 2^381 leaves 11 spare
// bits, so sums and differences are not reduced at all between multiplications.  Contracts (B(x) = bound of x in units of q):
//     mul(a, b)      needs B(a) * B(b) <= 2500  (a*b < 2^392 q)          -> result < 2q
//     add(a, b)      -> B(a) + B(b);    dbl(a) -> 2 B(a)
//     subk<J>(a, b)  needs B(b) <= 2^J                                     -> B(a) + 2^J      (a - b + 2^J q, limb-wise non-negative)
//     muladd(a,b,c,d) needs B(a) B(b) + B(c) B(d) <= 2500                 -> result < 2q  (a*b + c*d, one shared reduction)
//     wred(a)        needs B(a) <= 2000                                    -> < 4q            (top-limb quotient estimate)
//     is_zero(a)     needs B(a) <= 2000; exact test of a == 0 (mod q)
// Every point routine of zl_curve.h takes coordinates < 8q and returns coordinates < 8q; its comments carry the bound of every
// intermediate.  Memory / ABI formats are unchanged: load_* / store_* convert from / to arkworks' 32-bit-word layouts.
#pragma once
#include "zl_field.h"
#include "zl_mul28_gfx950.h"   // single-chain inline-asm product scans for 14 limbs (device only; gen_mul28.py)
#include "zl_mul28r_gfx950.h"  // ... and for 10 limbs

template <class P28, class P32>
struct alignas(16) Fp28 {
    static constexpr int L = P28::L;
    static constexpr int CANON_WORDS = P32::N;  // canonical / arkworks layouts: 32-bit words
    uint32_t l[L];
    uint32_t pad_[16 - L];  // 64-byte elements: an affine point is exactly one 128-B line

    ZL_HD static Fp28 zero() {
        Fp28 r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = 0;
#pragma unroll
        for (int i = 0; i < 16 - L; i++) r.pad_[i] = 0;
        return r;
    }
    ZL_HD static Fp28 one() {
        Fp28 r = zero();
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = P28::one(i);
        return r;
    }
    ZL_HD bool raw_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < L; i++) acc |= l[i];
        return acc == 0;
    }
    ZL_HD bool is_zero() const;  // == 0 mod q (value < 2000 q)
    ZL_HD bool operator==(const Fp28& o) const;
    ZL_HD bool operator!=(const Fp28& o) const { return !(*this == o); }
};

namespace zl {

template <class A, class B>
ZL_HD void carry28(Fp28<A, B>& r) {  // limbs < 2^32 - 2^4 -> limbs < 2^28 (value unchanged, top limb absorbs)
    constexpr int L = A::L;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        r.l[i + 1] += r.l[i] >> 28;
        r.l[i] &= 0xFFFFFFFu;
    }
}
template <class A, class B>
ZL_HD Fp28<A, B> add(const Fp28<A, B>& a, const Fp28<A, B>& b) {
    Fp28<A, B> r = a;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = a.l[i] + b.l[i];
    carry28(r);
    return r;
}
template <class A, class B>
ZL_HD Fp28<A, B> dbl(const Fp28<A, B>& a) {
    Fp28<A, B> r = a;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = a.l[i] << 1;
    carry28(r);
    return r;
}
// a - b + 2^J q, J in 1..6; needs b < 2^J q
template <int J, class A, class B>
ZL_HD Fp28<A, B> subk(const Fp28<A, B>& a, const Fp28<A, B>& b) {
    static_assert(J >= 1 && J <= 6, "bias table holds 2q .. 64q");
    Fp28<A, B> r = a;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = a.l[i] + A::kq(J, i) - b.l[i];  // limbs 0..12: >= 2^28 - b_i > 0; top limb: mod 2^32
    carry28(r);
    return r;
}
template <int J, class A, class B>
ZL_HD Fp28<A, B> negk(const Fp28<A, B>& a) {
    return subk<J>(Fp28<A, B>::zero(), a);
}
// Device forms of zl_curve.h's scan-only operands:
// a - b + 2^J q and 2^J q - b WITHOUT the carry pass ("fat": limbs 0..12 < 2^30), for an operand that ONLY feeds a product scan (never a
// squaring, a zero test, another subtraction or memory): the scans take any 32-bit limbs as long as their 64-bit columns hold, and a column
// of 28 products fat x carried (< 2^58 each) plus 14 of the reduction (< 2^56) stays below 2^63.  Needs a, b carried and b < 2^(J-1) q: the
// bias has one unit borrowed out of its top limb, so the top limb of 2^J q - b must be non-negative ON ITS OWN here (the carried forms
// let the carry pass absorb a wrap), which b < 2^(J-1) q guarantees (top limb of b <= that of 2^(J-1) q <= (that of 2^J q) / 2).
template <class A>
constexpr bool kq_limbs_biased() {
    for (int j = 1; j <= 6; j++) {
        for (int i = 0; i < A::L - 1; i++)
            if (A::kq(j, i) < (1u << 28) || A::kq(j, i) >= (1u << 29)) return false;
        if (j > 1 && A::kq(j, A::L - 1) + 1 < 2 * (A::kq(j - 1, A::L - 1) + 1)) return false;  // top limbs: that of 2^j q is at least twice that of 2^(j-1) q
    }
    return true;
}
template <int J, class A, class B>
ZL_HD Fp28<A, B> subk_scan(const Fp28<A, B>& a, const Fp28<A, B>& b) {
    static_assert(J >= 2 && J <= 6 && kq_limbs_biased<A>(), "bias table: limbs 0..12 of 2^j q in [2^28, 2^29)");
#if defined(__HIP_DEVICE_COMPILE__)
    Fp28<A, B> r = a;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = a.l[i] + A::kq(J, i) - b.l[i];
    return r;
#else
    return subk<J>(a, b);  // host: carried (the 56-bit fast path packs carried limbs)
#endif
}
template <int J, class A, class B>
ZL_HD Fp28<A, B> negk_scan(const Fp28<A, B>& b) {
    static_assert(J >= 2 && J <= 6 && kq_limbs_biased<A>(), "bias table: limbs 0..12 of 2^j q in [2^28, 2^29)");
#if defined(__HIP_DEVICE_COMPILE__)
    Fp28<A, B> r = b;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = A::kq(J, i) - b.l[i];
    return r;
#else
    return negk<J>(b);
#endif
}
// r^2 - ppp - 2q + 6q in ONE pass (device): the 6q bias is biased 2q + biased 4q with one more 2^28 borrowed from the next limb, so its limbs
// 0..12 lie in [3 * 2^28 - 1, 5 * 2^28) >= ppp_i + 2 q_i for carried ppp, q (scan outputs); the top limb is settled by the carry pass mod 2^32
template <class A, class B>
ZL_HD Fp28<A, B> x3_of(const Fp28<A, B>& rr, const Fp28<A, B>& ppp, const Fp28<A, B>& q) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int L = A::L;
    static_assert(kq_limbs_biased<A>(), "bias table");
    Fp28<A, B> x3 = rr;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const uint32_t b6 = A::kq(1, i) + A::kq(2, i) + (i < L - 1 ? (1u << 28) : 0u) - (i > 0 ? 1u : 0u);
        x3.l[i] = rr.l[i] + b6 - ppp.l[i] - 2u * q.l[i];  // < 2^28 + 5 * 2^28 < 2^31
    }
    carry28(x3);
    return x3;
#else
    return subk<2>(subk<1>(rr, ppp), dbl(q));
#endif
}
// generic spellings used by code that is shared with the 32-bit field (conservative biases)
template <class A, class B> ZL_HD Fp28<A, B> sub(const Fp28<A, B>& a, const Fp28<A, B>& b) { return subk<4>(a, b); }  // b < 16q
template <class A, class B> ZL_HD Fp28<A, B> neg(const Fp28<A, B>& a) { return negk<4>(a); }

// Montgomery product a*b/2^392 (+ multiple of q): < 2q when a*b < 2^392 q
template <class A, class B>
ZL_HD Fp28<A, B> mul_body28(const Fp28<A, B>& a, const Fp28<A, B>& b) {
    constexpr int L = A::L;
    uint32_t m[L];
    Fp28<A, B> r = a;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        m[k] = ((uint32_t)acc * A::INV) & 0xFFFFFFFu;
        acc += (uint64_t)m[k] * A::mod(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        r.l[k - L] = (uint32_t)acc & 0xFFFFFFFu;
        acc >>= 28;
    }
    return r;
}
// (a*b + c*d)/2^392 with ONE reduction: the two product scans share the column accumulators (<= 28 + 14 products of < 2^56 per
// column), so a difference of products costs 588 mads instead of 784.  < 2q when a*b + c*d < 2^392 q.
template <class A, class B>
ZL_HD Fp28<A, B> muladd_body28(const Fp28<A, B>& a, const Fp28<A, B>& b, const Fp28<A, B>& c, const Fp28<A, B>& d) {
    constexpr int L = A::L;
    uint32_t m[L];
    Fp28<A, B> r = a;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        m[k] = ((uint32_t)acc * A::INV) & 0xFFFFFFFu;
        acc += (uint64_t)m[k] * A::mod(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        r.l[k - L] = (uint32_t)acc & 0xFFFFFFFu;
        acc >>= 28;
    }
    return r;
}
// (a*b + c*d + e*f + g*h)/2^392 with one reduction (an Fq2 component of a sum of two Fq2 products): 784 + 196 mads.
// Columns hold <= 56 + 14 products of < 2^56 (top limbs slightly wider): < 2^63.  < 2q when the four products sum below 2^392 q.
template <class A, class B>
ZL_HD Fp28<A, B> muladd4_body28(const Fp28<A, B>& a, const Fp28<A, B>& b, const Fp28<A, B>& c, const Fp28<A, B>& d, const Fp28<A, B>& e,
                                const Fp28<A, B>& f, const Fp28<A, B>& g, const Fp28<A, B>& h) {
    constexpr int L = A::L;
    uint32_t m[L];
    Fp28<A, B> r = a;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
            acc += (uint64_t)e.l[i] * f.l[k - i];
            acc += (uint64_t)g.l[i] * h.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        m[k] = ((uint32_t)acc * A::INV) & 0xFFFFFFFu;
        acc += (uint64_t)m[k] * A::mod(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
            acc += (uint64_t)e.l[i] * f.l[k - i];
            acc += (uint64_t)g.l[i] * h.l[k - i];
        }
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * A::mod(k - i);
        r.l[k - L] = (uint32_t)acc & 0xFFFFFFFu;
        acc >>= 28;
    }
    return r;
}
// ---- host fast path (14 limbs): the same Montgomery product (R' = 2^392) on 7 limbs of 56 bits with 128-bit accumulators.  The host tails
// (window Horner of the plain path: 255 doublings per MSM, Groth16's scalar multiplications, normalisations) are 2-3x faster than on the
// 28-bit C++ scan; results are the same residues < 2q in the same limb layout.
#if !defined(__HIP_DEVICE_COMPILE__)
template <class A>
struct Host56 {
    static constexpr int H = A::L / 2;  // limbs of 56 bits (7 for BLS12-381, 5 for BN254)
    static constexpr uint64_t M56 = (1ull << 56) - 1;
    static uint64_t q(int j) { return (uint64_t)A::mod(2 * j) | ((uint64_t)A::mod(2 * j + 1) << 28); }
    static uint64_t inv() {  // -q^-1 mod 2^56 from the 28-bit constant by one Newton step
        const uint64_t q0 = q(0);
        uint64_t y = ((1ull << 28) - A::INV) & 0xFFFFFFFull;  // q^-1 mod 2^28
        y = (y * (2 - q0 * y)) & M56;
        return (0 - y) & M56;
    }
    template <class F>
    static void load(const F& a, uint64_t* o) {
        for (int j = 0; j < H; j++) o[j] = (uint64_t)a.l[2 * j] | ((uint64_t)a.l[2 * j + 1] << 28);  // top limb may exceed 28 bits: < 2^58 here
    }
    // r = (a*b [+ c*d]) / 2^392 mod q (+ multiple of q): product scan in base 2^56, one 128-bit column accumulator (a column holds at most
    // 7 + 7 products of < 2^116 and 7 of m q < 2^112: < 2^120), no per-product masking; SQ = a is b (28 distinct products instead of 49)
    template <bool CD, bool SQ, class F>
    static F mac_t(const F& a, const F& b, const F* c, const F* d) {
        typedef unsigned __int128 u128;
        static const uint64_t ninv = inv();
        uint64_t qa[H], x[H], y[H], u[H], v[H], m[H], t[H + 1];
#pragma unroll
        for (int j = 0; j < H; j++) qa[j] = q(j);
        load(a, x);
        if (!SQ) load(b, y);
        if (CD) { load(*c, u); load(*d, v); }
        u128 acc = 0;
#pragma unroll
        for (int k = 0; k < 2 * H; k++) {
            const int lo = k < H ? 0 : k - (H - 1), hi = k < H ? k : H - 1;
            if (SQ) {
                u128 cross = 0;
#pragma unroll
                for (int i = lo; 2 * i < k; i++) cross += (u128)x[i] * x[k - i];
                acc += cross + cross;
                if ((k & 1) == 0) acc += (u128)x[k >> 1] * x[k >> 1];
            } else {
#pragma unroll
                for (int i = lo; i <= hi; i++) acc += (u128)x[i] * y[k - i];
            }
            if (CD) {
#pragma unroll
                for (int i = lo; i <= hi; i++) acc += (u128)u[i] * v[k - i];
            }
            if (k < H) {
#pragma unroll
                for (int i = 0; i < k; i++) acc += (u128)m[i] * qa[k - i];
                m[k] = ((uint64_t)acc * ninv) & M56;
                acc += (u128)m[k] * qa[0];
            } else {
#pragma unroll
                for (int i = lo; i <= H - 1; i++) acc += (u128)m[i] * qa[k - i];
                t[k - H] = (uint64_t)acc & M56;
            }
            acc >>= 56;
        }
        t[H] = (uint64_t)acc;
        F r = a;
#pragma unroll
        for (int j = 0; j < H; j++) {
            r.l[2 * j] = (uint32_t)(t[j] & 0xFFFFFFFull);
            r.l[2 * j + 1] = (uint32_t)(t[j] >> 28);
        }
        r.l[A::L - 1] += (uint32_t)(t[H] << 28);  // zero for in-contract operands (result < 2q)
        return r;
    }
    template <class F>
    static F mac(const F& a, const F& b, const F* c, const F* d) {
        if (c) return mac_t<true, false>(a, b, c, d);
        if (&a == &b) return mac_t<false, true>(a, b, c, d);
        return mac_t<false, false>(a, b, c, d);
    }
};
#endif
template <class A, class B>
ZL_NOINLINE_HD Fp28<A, B> muladd_call28(Fp28<A, B> a, Fp28<A, B> b, Fp28<A, B> c, Fp28<A, B> d) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if constexpr (A::L % 2 == 0) return Host56<A>::mac(a, b, &c, &d);
#endif
    return muladd_body28(a, b, c, d);
}
// a*b + c*d (Montgomery), needs B(a) B(b) + B(c) B(d) <= 2500 -> < 2q
template <class A, class B>
ZL_HD Fp28<A, B> muladd(const Fp28<A, B>& a, const Fp28<A, B>& b, const Fp28<A, B>& c, const Fp28<A, B>& d) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_ASM_MUL28)
    if constexpr (A::L == 14) {
        Fp28<A, B> r = a;
        muladd28_asm<A>(r.l, a.l, b.l, c.l, d.l);
        return r;
    } else if constexpr (A::L == 10) {
        Fp28<A, B> r = a;
        muladd28r_asm<A>(r.l, a.l, b.l, c.l, d.l);
        return r;
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_INLINE_MUL28)
    return muladd_body28(a, b, c, d);
#else
    return muladd_call28<A, B>(a, b, c, d);
#endif
}
template <class A, class B>
ZL_NOINLINE_HD Fp28<A, B> mul_call28(Fp28<A, B> a, Fp28<A, B> b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    if constexpr (A::L % 2 == 0) return Host56<A>::mac(a, b, (const Fp28<A, B>*)nullptr, (const Fp28<A, B>*)nullptr);
#endif
    return mul_body28(a, b);
}
#if !defined(__HIP_DEVICE_COMPILE__)
template <class A, class B>
__attribute__((noinline)) Fp28<A, B> sqr_host28(const Fp28<A, B>& a) {
    return Host56<A>::mac(a, a, (const Fp28<A, B>*)nullptr, (const Fp28<A, B>*)nullptr);
}
#endif
template <class A, class B>
ZL_HD Fp28<A, B> mul(const Fp28<A, B>& a, const Fp28<A, B>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_ASM_MUL28)
    if constexpr (A::L == 14) {
        Fp28<A, B> r = a;
        mul28_asm<A>(r.l, a.l, b.l);
        return r;
    } else if constexpr (A::L == 10) {
        Fp28<A, B> r = a;
        mul28r_asm<A>(r.l, a.l, b.l);
        return r;
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_INLINE_MUL28)
    return mul_body28(a, b);  // device: inline (4 KB per site, a mixed addition stays inside the instruction cache)
#else
    return mul_call28<A, B>(a, b);
#endif
}
// a*b + c*d + e*f + g*h (Montgomery), needs the four bound products to sum to <= 2500 -> < 2q
template <class A, class B>
ZL_HD Fp28<A, B> muladd4(const Fp28<A, B>& a, const Fp28<A, B>& b, const Fp28<A, B>& c, const Fp28<A, B>& d, const Fp28<A, B>& e, const Fp28<A, B>& f,
                         const Fp28<A, B>& g, const Fp28<A, B>& h) {
    return muladd4_body28(a, b, c, d, e, f, g, h);
}
template <class A, class B>
ZL_HD Fp28<A, B> sqr(const Fp28<A, B>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_ASM_MUL28)
    if constexpr (A::L == 14) {
        Fp28<A, B> r = a;
        sqr28_asm<A>(r.l, a.l);
        return r;
    } else if constexpr (A::L == 10) {
        Fp28<A, B> r = a;
        sqr28r_asm<A>(r.l, a.l);
        return r;
    }
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
    if constexpr (A::L % 2 == 0) return sqr_host28<A, B>(a);
#endif
    return mul(a, a);
}
// weak reduction: value < 2000 q -> < 4q.  t = floor(top * floor(2^K / (qtop+1)) / 2^K) <= floor(a / q), short by <= 2 (a carried: the limb below the top < 2^28)
template <class A, class B>
ZL_HD Fp28<A, B> wred(const Fp28<A, B>& a) {
    constexpr int L = A::L;
    // quotient estimate from the top limb (BLS12-381: q >> 364 has 17 bits) or the top two (BN254: the top limb of q has 2 bits, q >> 224 has 30)
    uint64_t top = a.l[L - 1];
    if constexpr (A::QTOP_LIMBS == 2) top = (top << 28) + a.l[L - 2];
    const uint32_t t = (uint32_t)((top * A::QTOP_RECIP) >> A::QTOP_K);
    Fp28<A, B> r = a;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int64_t s = (int64_t)a.l[i] - (int64_t)((uint64_t)t * A::mod(i)) + c;
        r.l[i] = (uint32_t)s & 0xFFFFFFFu;
        c = s >> 28;
    }
    return r;  // the final carry is 0: a - t q >= 0 and < 2^392
}
// canonical representative in [0, q)
template <class A, class B>
ZL_HD Fp28<A, B> canon(const Fp28<A, B>& a) {
    constexpr int L = A::L;
    Fp28<A, B> r = wred(a);
#pragma unroll
    for (int rep = 0; rep < 3; rep++) {  // < 4q -> at most three subtractions
        uint32_t t[L];
        int64_t c = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            const int64_t s = (int64_t)r.l[i] - (int64_t)A::mod(i) + c;
            t[i] = (uint32_t)s & 0xFFFFFFFu;
            c = s >> 28;
        }
        const bool ge = c >= 0;  // no borrow out: r >= q
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = ge ? t[i] : r.l[i];
    }
    return r;
}
}  // namespace zl

template <class P28, class P32>
ZL_HD bool Fp28<P28, P32>::is_zero() const {
    const Fp28 r = zl::wred(*this);  // < 4q: zero mod q iff r in {0, q, 2q, 3q}
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (r.l[0] != P28::mq(k, 0)) continue;  // cheap filter on the low limb
        uint32_t diff = 0;
#pragma unroll
        for (int i = 0; i < L; i++) diff |= r.l[i] ^ P28::mq(k, i);
        any = any || diff == 0;
    }
    return any;
}
template <class P28, class P32>
ZL_HD bool Fp28<P28, P32>::operator==(const Fp28& o) const {
    return zl::subk<6>(zl::wred(*this), zl::wred(o)).is_zero();
}

namespace zl {
template <class A, class B>
ZL_HD Fp28<A, B> inv(const Fp28<A, B>& a) {  // Fermat a^(q-2); exponent from the 32-bit-word modulus
    constexpr int N = B::N;
    uint32_t e[N];
    uint32_t borrow = 2;
    for (int i = 0; i < N; i++) {
        const uint32_t m = B::mod(i);
        e[i] = m - borrow;
        borrow = m < borrow ? 1u : 0u;
    }
    Fp28<A, B> acc = Fp28<A, B>::one();
    for (int i = 32 * N - 1; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
}
// ---- layout conversions (32-bit little-endian words, as in the C ABI / arkworks) ---------------------------------------------
template <class A, class B>
ZL_HD Fp28<A, B> pack28(const uint32_t* w) {  // plain integer, CANON_WORDS 32-bit words -> 28-bit limbs
    Fp28<A, B> r = Fp28<A, B>::zero();
#pragma unroll
    for (int i = 0; i < A::L; i++) {
        const int bit = 28 * i, word = bit >> 5, sh = bit & 31;
        uint64_t v = word < B::N ? w[word] : 0;
        if (word + 1 < B::N) v |= (uint64_t)w[word + 1] << 32;
        r.l[i] = (uint32_t)(v >> sh) & 0xFFFFFFFu;
    }
    return r;
}
template <class A, class B>
ZL_HD void unpack28(uint32_t* w, const Fp28<A, B>& a) {  // limbs < 2^28, value < 2^(32*N)
#pragma unroll
    for (int j = 0; j < B::N; j++) w[j] = 0;
#pragma unroll
    for (int i = 0; i < A::L; i++) {
        const int bit = 28 * i, word = bit >> 5, sh = bit & 31;
        const uint64_t v = (uint64_t)a.l[i] << sh;
        if (word < B::N) w[word] |= (uint32_t)v;
        if (word + 1 < B::N) w[word + 1] |= (uint32_t)(v >> 32);
    }
}
#define ZL_CONST28(NAME, FN)                                  \
    template <class A, class B>                               \
    ZL_HD Fp28<A, B> NAME() {                                 \
        Fp28<A, B> r = Fp28<A, B>::zero();                    \
        _Pragma("unroll") for (int i = 0; i < A::L; i++) r.l[i] = A::FN(i); \
        return r;                                             \
    }
ZL_CONST28(const28_r2, r2)
ZL_CONST28(const28_from_m32, from_m32)
#undef ZL_CONST28
}  // namespace zl

// ---- quadratic extension helpers: Fq2 = Fq[u]/(u^2 + 1) on two Fp28 components --------------------------------------------------
// (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u: each component is ONE dual product scan (muladd_body28) with the
// subtraction done on a biased operand, so an Fq2 product is 1176 mads with two reductions and returns components < 2q, i.e. the same
// contract as the base field: the point formulas of zl_curve.h apply unchanged.  Operand components must be < 16q.
// On the device the three routines are real calls (the Fq2 point code would not fit the instruction cache inlined: a mixed
// addition is ~10.6 k mads); arguments travel as scalar words so that they stay in VGPRs (first 32) / the stack, never in a struct.
#define ZL_P14(x) uint32_t x##0, uint32_t x##1, uint32_t x##2, uint32_t x##3, uint32_t x##4, uint32_t x##5, uint32_t x##6, uint32_t x##7, uint32_t x##8, uint32_t x##9, uint32_t x##10, uint32_t x##11, uint32_t x##12, uint32_t x##13
#define ZL_A14(v) v.l[0], v.l[1], v.l[2], v.l[3], v.l[4], v.l[5], v.l[6], v.l[7], v.l[8], v.l[9], v.l[10], v.l[11], v.l[12], v.l[13]
#define ZL_S14(v, x) v.l[0] = x##0; v.l[1] = x##1; v.l[2] = x##2; v.l[3] = x##3; v.l[4] = x##4; v.l[5] = x##5; v.l[6] = x##6; v.l[7] = x##7; v.l[8] = x##8; v.l[9] = x##9; v.l[10] = x##10; v.l[11] = x##11; v.l[12] = x##12; v.l[13] = x##13
// ... and the 10-limb instance (BN254 G2, round 4)
#define ZL_P10(x) uint32_t x##0, uint32_t x##1, uint32_t x##2, uint32_t x##3, uint32_t x##4, uint32_t x##5, uint32_t x##6, uint32_t x##7, uint32_t x##8, uint32_t x##9
#define ZL_A10(v) v.l[0], v.l[1], v.l[2], v.l[3], v.l[4], v.l[5], v.l[6], v.l[7], v.l[8], v.l[9]
#define ZL_S10(v, x) v.l[0] = x##0; v.l[1] = x##1; v.l[2] = x##2; v.l[3] = x##3; v.l[4] = x##4; v.l[5] = x##5; v.l[6] = x##6; v.l[7] = x##7; v.l[8] = x##8; v.l[9] = x##9
template <int L> struct PairL { uint32_t l[2 * L]; };  // two L-limb components, returned in registers
using Pair28 = PairL<14>;
namespace zl {
template <class A, class B>
ZL_HD PairL<A::L> pair28(const Fp28<A, B>& c0, const Fp28<A, B>& c1) {
    PairL<A::L> r;
#pragma unroll
    for (int i = 0; i < A::L; i++) { r.l[i] = c0.l[i]; r.l[A::L + i] = c1.l[i]; }
    return r;
}
template <class A, class B>
ZL_HD void unpair28(const PairL<A::L>& p, Fp28<A, B>& c0, Fp28<A, B>& c1) {
    c0 = Fp28<A, B>::zero();
    c1 = Fp28<A, B>::zero();
#pragma unroll
    for (int i = 0; i < A::L; i++) { c0.l[i] = p.l[i]; c1.l[i] = p.l[A::L + i]; }
}
template <class A, class B>
ZL_HD PairL<A::L> fq2_mul_body28(const Fp28<A, B>& a0, const Fp28<A, B>& a1, const Fp28<A, B>& b0, const Fp28<A, B>& b1) {  // (a + b u)(c + d u), components < 16q
    const Fp28<A, B> nb1 = negk_scan<5>(b1);  // scan-only operand: 32q - b1, un-carried on the device (b1 <= 16q)
    return pair28(muladd(a0, b0, a1, nb1), muladd(a0, b1, a1, b0));
}
template <class A, class B>
ZL_HD PairL<A::L> fq2_sqr_body28(const Fp28<A, B>& a0, const Fp28<A, B>& a1) {  // (a + b u)^2 = (a + b)(a - b) + 2ab u, components < 16q
    return pair28(mul(add(a0, a1), subk<4>(a0, a1)), mul(dbl(a0), a1));
}
template <class A, class B>
ZL_NOINLINE_HD Pair28 fq2_mul_call28(ZL_P14(wa), ZL_P14(wb), ZL_P14(wc), ZL_P14(wd)) {
    static_assert(A::L == 14, "the 14-limb entry");
    Fp28<A, B> a0 = Fp28<A, B>::zero(), a1 = a0, b0 = a0, b1 = a0;
    ZL_S14(a0, wa); ZL_S14(a1, wb); ZL_S14(b0, wc); ZL_S14(b1, wd);
    return fq2_mul_body28(a0, a1, b0, b1);
}
template <class A, class B>
ZL_NOINLINE_HD Pair28 fq2_sqr_call28(ZL_P14(wa), ZL_P14(wb)) {
    static_assert(A::L == 14, "the 14-limb entry");
    Fp28<A, B> a0 = Fp28<A, B>::zero(), a1 = a0;
    ZL_S14(a0, wa); ZL_S14(a1, wb);
    return fq2_sqr_body28(a0, a1);
}
template <class A, class B>
ZL_NOINLINE_HD PairL<10> fq2_mul_call28x10(ZL_P10(wa), ZL_P10(wb), ZL_P10(wc), ZL_P10(wd)) {
    static_assert(A::L == 10, "the 10-limb entry");
    Fp28<A, B> a0 = Fp28<A, B>::zero(), a1 = a0, b0 = a0, b1 = a0;
    ZL_S10(a0, wa); ZL_S10(a1, wb); ZL_S10(b0, wc); ZL_S10(b1, wd);
    return fq2_mul_body28(a0, a1, b0, b1);
}
template <class A, class B>
ZL_NOINLINE_HD PairL<10> fq2_sqr_call28x10(ZL_P10(wa), ZL_P10(wb)) {
    static_assert(A::L == 10, "the 10-limb entry");
    Fp28<A, B> a0 = Fp28<A, B>::zero(), a1 = a0;
    ZL_S10(a0, wa); ZL_S10(a1, wb);
    return fq2_sqr_body28(a0, a1);
}
}  // namespace zl

// ---- uniform conversion API for both field representations (used at every memory / ABI boundary) -----------------------------
template <class F>
struct FieldIO;
template <class P>
struct FieldIO<Fp<P>> {
    static constexpr int WORDS = P::N;
    ZL_HD static Fp<P> load_canon(const uint32_t* w) { Fp<P> c; for (int i = 0; i < P::N; i++) c.l[i] = w[i]; return zl::to_mont(c); }
    ZL_HD static Fp<P> load_mont32(const uint32_t* w) { Fp<P> c; for (int i = 0; i < P::N; i++) c.l[i] = w[i]; return c; }
    ZL_HD static void store_canon(uint32_t* w, const Fp<P>& a) { const Fp<P> c = zl::from_mont(a); for (int i = 0; i < P::N; i++) w[i] = c.l[i]; }
};
template <class A, class B>
struct FieldIO<Fp28<A, B>> {
    using F = Fp28<A, B>;
    static constexpr int WORDS = B::N;
    ZL_HD static F load_canon(const uint32_t* w) { return zl::canon(zl::mul(zl::pack28<A, B>(w), zl::const28_r2<A, B>())); }
    ZL_HD static F load_mont32(const uint32_t* w) { return zl::canon(zl::mul(zl::pack28<A, B>(w), zl::const28_from_m32<A, B>())); }
    ZL_HD static void store_canon(uint32_t* w, const F& a) {
        F one = F::zero();
        one.l[0] = 1;
        zl::unpack28(w, zl::canon(zl::mul(zl::wred(a), one)));
    }
};
