// zl_field28r.h -- the scalar fields (BLS12-381 Fr, BN254 Fr) on 10 unsaturated 28-bit limbs with lazy reduction, for the NTT passes (round 4;
// VERDICT r3 item 8).
//
// Why: the 8 x 32-bit Fr arithmetic pays a carry chain in every product column and a compare + conditional subtraction in every butterfly addition
// (profiles/r04_fbench_f64.log, live data, 3 waves / SIMD: 108 G products/s, 81.7 G butterflies/s).  On 28-bit limbs a product is a carry-free chain of
// 210 v_mad_u64_u32 (113 G/s) and a butterfly addition is ten v_add_u32 + one carry pass with NO comparison: 97.8 G butterflies/s (x 1.20).
//
// Representation: value = sum l_i 2^(28 i); limbs 0..8 carried (< 2^28), the top limb holds the rest; the VALUE is only bounded (B r), never compared
// with r between multiplications.  The Montgomery radix R' = 2^280 belongs to the MULTIPLIER alone: mul(x, w R') = x w, so the data keeps the
// form the caller gave it (canonical integers or R = 2^256 Montgomery residues) and only the twiddle tables are stored as w R' mod r.
// Contracts (B(x) = bound of x in units of r):
//     mul(a, b)        needs B(a) B(b) < 2^25            -> < 2r   ((a b + m r) / R' with m < R': < a b / R' + r)
//     add(a, b)        -> B(a) + B(b)
//     subk(a, b, K)    K = biased limbs of 2^j r with 2^j >= B(b) + 2 (P::kq(j, .))   -> B(a) + 2^j
//     wred(a)          needs B(a) < 2^20                  -> < 2r   (quotient estimate from the top limb)
//     canon(a)         needs B(a) < 2^20                  -> the canonical residue < r
//     pack(a)          needs a carried and < 2^256 (any a < 2r)
// Everything is ZL_HD (the host path exists for the unit test of this header, tests/test_fr28.py; the product only uses it on the device).
#pragma once
#include "zl_field.h"
#include "zl_mul28r_gfx950.h"  // single-chain inline-asm product scan for 10 limbs of 28 bits (device only; gen_mul28.py)
#include "zl_mul29r_gfx950.h"  // ... and for 9 limbs of 29 bits (round 5)

// Round 5: the header is generic in the limb count L = P::L and width BITS = P::BITS of its parameter struct (zl_params.h: *_Fr28 = 10 x 28, R' = 2^280, and
// *_Fr29 = 9 x 29, R' = 2^261).  The 29-bit instance trades spare bits for work: a product is 162 + 9 mads instead of 200 + 10, an addition 9 limbs instead of 10,
// but mul(a, b) only takes B(a) B(b) <= P::MUL_BOUND = floor(R' / r) = 70 (BLS12-381) / 169 (BN254) instead of 2^25, so a chain of additions must be weakly
// reduced (wred: ~30 instructions against ~190 of a product) before it exceeds that.  "28" in the names below is historical.
template <class P>
struct Fr28 {
    static constexpr int L = P::L;
    static constexpr int BITS = P::BITS;
    static constexpr uint32_t MASK = (1u << P::BITS) - 1u;
    uint32_t l[P::L];
};

namespace zl {

// 8 x 32-bit words (value < 2^256) -> L carried limbs
template <class P>
ZL_HD Fr28<P> unpack28r(const uint32_t* __restrict__ w) {
    constexpr int L = P::L, B = P::BITS;
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int bit = B * i, wd = bit >> 5, sh = bit & 31;
        uint32_t v = w[wd] >> sh;
        if (sh > 32 - B && wd + 1 < 8) v |= w[wd + 1] << (32 - sh);
        r.l[i] = v & Fr28<P>::MASK;
    }
    return r;  // the top limb holds what is left of the 256 bits (4 bits at 10 x 28, 24 bits at 9 x 29)
}
// carried limbs of a value < 2^256 -> 8 x 32-bit words
template <class P>
ZL_HD void pack28r(uint32_t* __restrict__ w, const Fr28<P>& a) {
    constexpr int L = P::L, B = P::BITS;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, li = bit / B, sh = bit % B;  // word j starts inside limb li at bit sh
        uint32_t v = a.l[li] >> sh;
        if (li + 1 < L) v |= a.l[li + 1] << (B - sh);
        if (2 * B - sh < 32 && li + 2 < L) v |= a.l[li + 2] << (2 * B - sh);
        w[j] = v;
    }
}
template <class P>
ZL_HD void carry28r(Fr28<P>& r) {  // limbs < 2^32 - 2^(32 - BITS) -> limbs 0..L-2 < 2^BITS (value unchanged, the top limb absorbs)
#pragma unroll
    for (int i = 0; i < P::L - 1; i++) {
        r.l[i + 1] += r.l[i] >> P::BITS;
        r.l[i] &= Fr28<P>::MASK;
    }
}
template <class P>
ZL_HD Fr28<P> add(const Fr28<P>& a, const Fr28<P>& b) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = a.l[i] + b.l[i];
    carry28r(r);
    return r;
}
// The same without the carry pass ("fat" limbs, < 2^31): for a value that only feeds a product (the scan takes any limbs whose 64-bit columns hold: L
// products of fat x carried -- < 2^59 each at 28 bits, < 2^60 at 29 -- plus the reduction's L products < 2^(2 BITS): below 2^63.5 in both instances), an
// addition that is carried afterwards, or the MINUEND of a biased subtraction.
template <class P>
ZL_HD Fr28<P> add_nc(const Fr28<P>& a, const Fr28<P>& b) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class P>
ZL_HD Fr28<P> subk_nc(const Fr28<P>& a, const Fr28<P>& b, const uint32_t* __restrict__ K) {  // b carried (limbs 0..L-2 below 2^BITS)
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = a.l[i] + K[i] - b.l[i];
    return r;
}
// a - b + K, K = the biased limbs of 2^j r (limbs 0..L-2 in [2^BITS, 2^(BITS+1)), so no limb of a + K - b is negative for carried b; top limb: 2^j >= B(b) + 2)
template <class P>
ZL_HD Fr28<P> subk(const Fr28<P>& a, const Fr28<P>& b, const uint32_t* __restrict__ K) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = a.l[i] + K[i] - b.l[i];
    carry28r(r);
    return r;
}
template <class P>
ZL_HD Fr28<P> mul_body28r(const Fr28<P>& a, const Fr28<P>& b) {  // plain product scan (host; device fallback)
    constexpr int L = P::L, B = P::BITS;
    uint32_t m[L];
    Fr28<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::mod(k - i);
        m[k] = ((uint32_t)acc * P::INV) & Fr28<P>::MASK;
        acc += (uint64_t)m[k] * P::mod(0);
        acc >>= B;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::mod(k - i);
        r.l[k - L] = (uint32_t)acc & Fr28<P>::MASK;
        acc >>= B;
    }
    return r;
}
template <class P>
ZL_HD Fr28<P> mul(const Fr28<P>& a, const Fr28<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_ASM_MUL28)
    Fr28<P> r = a;
    if constexpr (P::L == 10 && P::BITS == 28) mul28r_asm<P>(r.l, a.l, b.l);
    else if constexpr (P::L == 9 && P::BITS == 29) mul29r_asm<P>(r.l, a.l, b.l);
    else r = mul_body28r(a, b);
    return r;
#else
    return mul_body28r(a, b);
#endif
}
// weak reduction of a CARRIED a: -> < 2r.  t = the top limb = floor(a / 2^T), T = BITS (L - 1); qh = floor(t MU / 2^MU_SHIFT) with MU = floor(2^(T + MU_SHIFT) / r)
// never exceeds floor(a / r) and falls short of a / r by less than 2^T / r + t / 2^MU_SHIFT + 1 < 1.2 (t < 2^32, MU_SHIFT >= 32; 2^T / r < 2^-2), so
// 0 <= a - qh r < 2r.  Needs B(a) < 2^20 (28 bits: the top limb below 2^32 - 2^29) resp. B(a) <= 128 (29 bits: a < 2^262, top limb < 2^30).
template <class P>
ZL_HD Fr28<P> wred(const Fr28<P>& a) {
    constexpr int L = P::L, B = P::BITS;
    const uint32_t qh = (uint32_t)(((uint64_t)a.l[L - 1] * P::MU) >> P::MU_SHIFT);
    Fr28<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int64_t s = (int64_t)a.l[i] - (int64_t)((uint64_t)qh * P::mod(i)) + c;
        r.l[i] = i < L - 1 ? ((uint32_t)s & Fr28<P>::MASK) : (uint32_t)s;
        c = s >> B;
    }
    return r;
}
// the canonical residue: weak reduction, then one conditional subtraction of r
template <class P>
ZL_HD Fr28<P> canon(const Fr28<P>& a) {
    constexpr int L = P::L, B = P::BITS;
    const Fr28<P> x = wred(a);
    Fr28<P> d;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int64_t s = (int64_t)x.l[i] - (int64_t)P::mod(i) + c;
        d.l[i] = i < L - 1 ? ((uint32_t)s & Fr28<P>::MASK) : (uint32_t)s;
        c = s >> B;
    }
    const bool ge = c >= 0;  // no borrow out of the top limb: x >= r
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = ge ? d.l[i] : x.l[i];
    return r;
}

}  // namespace zl
