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
#include "zl_mul28r_gfx950.h"  // single-chain inline-asm product scan for 10 limbs (device only; gen_mul28.py)

template <class P>
struct Fr28 {
    static constexpr int L = 10;
    uint32_t l[10];
};

namespace zl {

// 8 x 32-bit words (value < 2^256) -> 10 carried limbs
template <class P>
ZL_HD Fr28<P> unpack28r(const uint32_t* __restrict__ w) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int bit = 28 * i, wd = bit >> 5, sh = bit & 31;
        uint32_t v = w[wd] >> sh;
        if (sh > 4 && wd + 1 < 8) v |= w[wd + 1] << (32 - sh);
        r.l[i] = v & 0xFFFFFFFu;
    }
    return r;  // limb 9 = bits 252..255: 4 bits
}
// carried limbs of a value < 2^256 -> 8 x 32-bit words
template <class P>
ZL_HD void pack28r(uint32_t* __restrict__ w, const Fr28<P>& a) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, li = bit / 28, sh = bit % 28;  // word j starts inside limb li at bit sh
        uint32_t v = a.l[li] >> sh;
        v |= a.l[li + 1] << (28 - sh);                          // li + 1 <= 9: word 7 starts at limb 8 (bit 224 = 8 * 28)
        if (28 - sh + 28 < 32 && li + 2 < 10) v |= a.l[li + 2] << (56 - sh);
        w[j] = v;
    }
}
template <class P>
ZL_HD void carry28r(Fr28<P>& r) {  // limbs < 2^32 - 2^4 -> limbs 0..8 < 2^28 (value unchanged, the top limb absorbs)
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i + 1] += r.l[i] >> 28;
        r.l[i] &= 0xFFFFFFFu;
    }
}
template <class P>
ZL_HD Fr28<P> add(const Fr28<P>& a, const Fr28<P>& b) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + b.l[i];
    carry28r(r);
    return r;
}
// The same without the carry pass ("fat" limbs, < 2^31): for a value that only feeds a product (the scan takes any limbs whose 64-bit columns hold: ten
// products of fat x carried, < 2^59 each, plus the reduction's), an addition that is carried afterwards, or the MINUEND of a biased subtraction.
template <class P>
ZL_HD Fr28<P> add_nc(const Fr28<P>& a, const Fr28<P>& b) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <class P>
ZL_HD Fr28<P> subk_nc(const Fr28<P>& a, const Fr28<P>& b, const uint32_t* __restrict__ K) {  // b carried (limbs 0..8 below 2^28)
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + K[i] - b.l[i];
    return r;
}
// a - b + K, K = the ten biased limbs of 2^j r (limbs 0..8 in [2^28, 2^29), so no limb of a + K - b is negative for carried b; top limb: 2^j >= B(b) + 2)
template <class P>
ZL_HD Fr28<P> subk(const Fr28<P>& a, const Fr28<P>& b, const uint32_t* __restrict__ K) {
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + K[i] - b.l[i];
    carry28r(r);
    return r;
}
template <class P>
ZL_HD Fr28<P> mul_body28r(const Fr28<P>& a, const Fr28<P>& b) {  // plain product scan (host; device fallback)
    uint32_t m[10];
    Fr28<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::mod(k - i);
        m[k] = ((uint32_t)acc * P::INV) & 0xFFFFFFFu;
        acc += (uint64_t)m[k] * P::mod(0);
        acc >>= 28;
    }
#pragma unroll
    for (int k = 10; k < 20; k++) {
#pragma unroll
        for (int i = k - 9; i < 10; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 9; i < 10; i++) acc += (uint64_t)m[i] * P::mod(k - i);
        r.l[k - 10] = (uint32_t)acc & 0xFFFFFFFu;
        acc >>= 28;
    }
    return r;
}
template <class P>
ZL_HD Fr28<P> mul(const Fr28<P>& a, const Fr28<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_NO_ASM_MUL28)
    Fr28<P> r = a;
    mul28r_asm<P>(r.l, a.l, b.l);
    return r;
#else
    return mul_body28r(a, b);
#endif
}
// weak reduction: B(a) < 2^20 -> < 2r.  t = floor(a / 2^252) is the top limb; qh = floor(t MU / 2^32) with MU = floor(2^284 / r) never exceeds
// floor(a / r) and falls short of a / r by less than 2^252 / r + 2^-12 t / ... < 1.2, so 0 <= a - qh r < 2r (exhaustive over t in the host test).
template <class P>
ZL_HD Fr28<P> wred(const Fr28<P>& a) {
    const uint32_t qh = (uint32_t)(((uint64_t)a.l[9] * P::MU) >> 32);
    Fr28<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int64_t s = (int64_t)a.l[i] - (int64_t)((uint64_t)qh * P::mod(i)) + c;
        r.l[i] = i < 9 ? ((uint32_t)s & 0xFFFFFFFu) : (uint32_t)s;
        c = s >> 28;
    }
    return r;
}
// the canonical residue: weak reduction, then one conditional subtraction of r
template <class P>
ZL_HD Fr28<P> canon(const Fr28<P>& a) {
    const Fr28<P> x = wred(a);
    Fr28<P> d;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int64_t s = (int64_t)x.l[i] - (int64_t)P::mod(i) + c;
        d.l[i] = i < 9 ? ((uint32_t)s & 0xFFFFFFFu) : (uint32_t)s;
        c = s >> 28;
    }
    const bool ge = c >= 0;  // no borrow out of the top limb: x >= r
    Fr28<P> r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = ge ? d.l[i] : x.l[i];
    return r;
}

}  // namespace zl
