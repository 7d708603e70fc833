// zl_msm_endo.h -- endomorphism front ends of the MSM: GLV on BLS12-381 / BN254 G1 (k = k1 + k2 lambda) and GLS on BLS12-381 G2 (four base-|z|
// digits over P, psi(P), psi^2(P), psi^3(P)): scalar splits and the images of the bases.  Instantiated where they are launched (zl_msm.hip).
#pragma once
#include "zl_ctx.h"
#include "zl_msm_common.h"

// ------------------------------------------------------------------------------------------------ GLV front end
// BLS12-381 G1 has the endomorphism phi(x, y) = (beta x, y) = [lambda](x, y) with lambda = z^2 - 1 and r = lambda^2 + lambda + 1.  A plain
// MSM over n points and 255-bit scalars becomes one over 2n points (P_i and phi(P_i)) and signed 127-bit half-scalars:
//     k = k1 + k2 lambda,  k2 = floor(k / lambda), k1 = k mod lambda,  then balanced into |k1|, |k2| <= lambda / 2 + 1 < 2^127
//     (k1 > lambda / 2: k1 -= lambda, k2 += 1;   k2 > lambda / 2: k2 -= lambda + 1, k1 -= 1   -- lambda^2 = -lambda - 1 mod r)
// The number of (point, window) additions is unchanged (2n half-scalars x half as many windows), but there are half as many bucket
// sets to merge and reduce and half as many windows in the host Horner -- the parts that dominate small and mid-size MSMs.  The group
// law makes the result identical.  arkworks 0.3 does not use the endomorphism in VariableBaseMSM; results do not depend on it.
struct zl_u128 { uint64_t lo, hi; };
__device__ __forceinline__ bool zl_gt(zl_u128 a, zl_u128 b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
__device__ __forceinline__ bool zl_ge(zl_u128 a, zl_u128 b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
__device__ __forceinline__ zl_u128 zl_sub(zl_u128 a, zl_u128 b) { return zl_u128{a.lo - b.lo, a.hi - b.hi - (a.lo < b.lo ? 1u : 0u)}; }
__device__ __forceinline__ zl_u128 zl_inc(zl_u128 a) { return zl_u128{a.lo + 1, a.hi + (a.lo + 1 == 0 ? 1u : 0u)}; }
__device__ __forceinline__ zl_u128 zl_dec(zl_u128 a) { return zl_u128{a.lo - 1, a.hi - (a.lo == 0 ? 1u : 0u)}; }
// Scalars in [r, 2^SC_BITS) pass zl_flag_wide_scalar but are not canonical: floor(k / lambda) then exceeds lambda + 1 and the balanced halves wrap.  The
// plain path returns the sum mod r for them, so the endomorphism splits reduce such a scalar once (k < 2^255 < 2 r) and return the same point.
template <class P>
__device__ __forceinline__ void zl_reduce_once_mod_r(uint32_t* k) {
    uint32_t d[8];
    uint32_t borrow = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const uint64_t x = (uint64_t)k[w] - P::rmod(w) - borrow;
        d[w] = (uint32_t)x;
        borrow = (uint32_t)(x >> 63);
    }
    if (!borrow) {
#pragma unroll
        for (int w = 0; w < 8; w++) k[w] = d[w];
    }
}
// out: 2n records of 8 words -- record i = k1 of scalar i, record n + i = k2 of scalar i; magnitude in words 0..3, sign in bit 31 of word 7.
// Scalars of bases at infinity give two zero records; a scalar with bits at or above sc_bits sets *bad (not canonical).
template <class P>
__global__ void __launch_bounds__(256) k_glv_split(const uint32_t* __restrict__ scalars, uint32_t n, const uint8_t* __restrict__ inf, uint32_t* __restrict__ out,
                                                    int sc_bits, uint32_t* __restrict__ bad) {
    ZL_SIDE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 lo4 = sp[0], hi4 = sp[1];
    zl_flag_wide_scalar(hi4.w, sc_bits, bad);
    if (inf && inf[i]) lo4 = hi4 = make_uint4(0, 0, 0, 0);
    uint32_t k[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    zl_reduce_once_mod_r<P>(k);
    // q = floor(k m / 2^383), m = floor(2^383 / lambda): the quotient or one less
    uint32_t pw[16];
    {
        uint64_t acc = 0;
        uint32_t top = 0;
#pragma unroll
        for (int col = 0; col < 15; col++) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const int b = col - a;
                if (b < 0 || b > 7) continue;
                const uint64_t pr = (uint64_t)k[a] * P::barrett(b);
                acc += pr;
                top += acc < pr ? 1u : 0u;
            }
            pw[col] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)top << 32);
            top = 0;
        }
        pw[15] = (uint32_t)acc;
    }
    zl_u128 q{((uint64_t)(pw[11] >> 31) | ((uint64_t)pw[12] << 1) | ((uint64_t)pw[13] << 33)), ((uint64_t)(pw[13] >> 31) | ((uint64_t)pw[14] << 1) | ((uint64_t)pw[15] << 33))};
    const uint32_t qw[4] = {(uint32_t)q.lo, (uint32_t)(q.lo >> 32), (uint32_t)q.hi, (uint32_t)(q.hi >> 32)};
    // k1 = k - q lambda (mod 2^160; the true value is below 2 lambda < 2^129)
    uint32_t t[5];
    {
        uint64_t acc = 0;
        uint32_t top = 0;
#pragma unroll
        for (int col = 0; col < 5; col++) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int b = col - a;
                if (b < 0 || b > 3) continue;
                const uint64_t pr = (uint64_t)qw[a] * P::lambda(b);
                acc += pr;
                top += acc < pr ? 1u : 0u;
            }
            t[col] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)top << 32);
            top = 0;
        }
    }
    uint32_t d[5];
    {
        uint32_t borrow = 0;
#pragma unroll
        for (int w = 0; w < 5; w++) {
            const uint64_t x = (uint64_t)k[w] - t[w] - borrow;
            d[w] = (uint32_t)x;
            borrow = (uint32_t)(x >> 63);
        }
    }
    const zl_u128 LAM{(uint64_t)P::lambda(0) | ((uint64_t)P::lambda(1) << 32), (uint64_t)P::lambda(2) | ((uint64_t)P::lambda(3) << 32)};
    const zl_u128 HALF{(LAM.lo >> 1) | (LAM.hi << 63), LAM.hi >> 1};
    zl_u128 k1{(uint64_t)d[0] | ((uint64_t)d[1] << 32), (uint64_t)d[2] | ((uint64_t)d[3] << 32)};
    uint32_t k1top = d[4];
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        if (k1top != 0u || zl_ge(k1, LAM)) {
            const bool br = zl_gt(LAM, k1);
            k1 = zl_sub(k1, LAM);
            k1top -= br ? 1u : 0u;
            q = zl_inc(q);
        }
    }
    zl_u128 k2 = q;
    uint32_t neg1 = 0, neg2 = 0;
    if (zl_gt(k1, HALF)) { k1 = zl_sub(LAM, k1); neg1 = 1; k2 = zl_inc(k2); }
    if (zl_gt(k2, HALF)) {
        k2 = zl_sub(zl_inc(LAM), k2);
        neg2 = 1;
        if (neg1) k1 = zl_inc(k1);
        else if ((k1.lo | k1.hi) == 0) { k1.lo = 1; neg1 = 1; }
        else k1 = zl_dec(k1);
    }
    if ((k1.lo | k1.hi) == 0) neg1 = 0;
    if ((k2.lo | k2.hi) == 0) neg2 = 0;
    uint4* o1 = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    uint4* o2 = reinterpret_cast<uint4*>(out + ((size_t)n + i) * 8);
    o1[0] = make_uint4((uint32_t)k1.lo, (uint32_t)(k1.lo >> 32), (uint32_t)k1.hi, (uint32_t)(k1.hi >> 32));
    o1[1] = make_uint4(0, 0, 0, neg1 << 31);
    o2[0] = make_uint4((uint32_t)k2.lo, (uint32_t)(k2.lo >> 32), (uint32_t)k2.hi, (uint32_t)(k2.hi >> 32));
    o2[1] = make_uint4(0, 0, 0, neg2 << 31);
}
// BN254 G1: r is not lambda^2 + lambda + 1 with a 128-bit lambda, so the split is the two-dimensional one of GLV (2001): with the short basis v1 = (a1, b1), v2 = (a2, b2) of the
// lattice {(a, b): a + b lambda = 0 mod r}, c1 = round(b2 k / r), c2 = round(-b1 k / r) and (k1, k2) = (k, 0) - c1 v1 - c2 v2.  The roundings are taken as
// (k g_i + 2^255) >> 256 with g_i = round(2^256 |.| / r): at most one off, which moves (k1, k2) by one basis vector and keeps k1 + k2 lambda = k (mod r) exactly.
// All products are taken on magnitudes, modulo 2^192; |k1|, |k2| < 0.6 * 2^127 (tools/gen_bn254_glv.py runs this arithmetic on edge and random scalars).
// Same record format as k_glv_split.
template <int NA, int NB, int NO>
__device__ __forceinline__ void zl_mul_words(const uint32_t* a, const uint32_t* b, uint32_t* out) {  // out[0 .. NO) = low NO words of a[0 .. NA) * b[0 .. NB)
    uint64_t acc = 0;
    uint32_t top = 0;
#pragma unroll
    for (int col = 0; col < NO; col++) {
#pragma unroll
        for (int x = 0; x < NA; x++) {
            const int y = col - x;
            if (y < 0 || y >= NB) continue;
            const uint64_t pr = (uint64_t)a[x] * b[y];
            acc += pr;
            top += acc < pr ? 1u : 0u;
        }
        out[col] = (uint32_t)acc;
        acc = (acc >> 32) | ((uint64_t)top << 32);
        top = 0;
    }
}
template <int N>
__device__ __forceinline__ void zl_addsub_words(uint32_t* acc, const uint32_t* v, bool sub) {  // acc +-= v (mod 2^(32 N))
    uint32_t carry = sub ? 1u : 0u;  // a - v = a + ~v + 1
#pragma unroll
    for (int w = 0; w < N; w++) {
        const uint64_t x = (uint64_t)acc[w] + (sub ? ~v[w] : v[w]) + carry;
        acc[w] = (uint32_t)x;
        carry = (uint32_t)(x >> 32);
    }
}
template <class P>
__global__ void __launch_bounds__(256) k_glv_split_lattice(const uint32_t* __restrict__ scalars, uint32_t n, const uint8_t* __restrict__ inf, uint32_t* __restrict__ out,
                                                            int sc_bits, uint32_t* __restrict__ bad) {
    ZL_SIDE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 lo4 = sp[0], hi4 = sp[1];
    zl_flag_wide_scalar(hi4.w, sc_bits, bad);
    if (inf && inf[i]) lo4 = hi4 = make_uint4(0, 0, 0, 0);
    uint32_t k[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    zl_reduce_once_mod_r<P>(k);
    uint32_t g1[5], g2[5], a1[4], b1[4], a2[4], b2[4];
#pragma unroll
    for (int w = 0; w < 5; w++) { g1[w] = P::g1(w); g2[w] = P::g2(w); }
#pragma unroll
    for (int w = 0; w < 4; w++) { a1[w] = P::a1(w); b1[w] = P::b1(w); a2[w] = P::a2(w); b2[w] = P::b2(w); }
    // c_i = (k g_i + 2^255) >> 256: words 8 .. 11 of the 13-word product (c1 < 2^66, c2 < 2^127)
    uint32_t pr[13], c1[4], c2[4];
    const uint32_t half[8] = {0, 0, 0, 0, 0, 0, 0, 0x80000000u};
    zl_mul_words<8, 5, 13>(k, g1, pr);
    {
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < 13; w++) {
            const uint64_t x = (uint64_t)pr[w] + (w < 8 ? half[w] : 0u) + carry;
            pr[w] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
    }
#pragma unroll
    for (int w = 0; w < 4; w++) c1[w] = pr[8 + w];
    zl_mul_words<8, 5, 13>(k, g2, pr);
    {
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < 13; w++) {
            const uint64_t x = (uint64_t)pr[w] + (w < 8 ? half[w] : 0u) + carry;
            pr[w] = (uint32_t)x;
            carry = (uint32_t)(x >> 32);
        }
    }
#pragma unroll
    for (int w = 0; w < 4; w++) c2[w] = pr[8 + w];
    // k1 = k -+ c1 |a1| -+ c2 |a2|,  k2 = -+ c1 |b1| -+ c2 |b2|   (mod 2^192; the true values are below 2^127 in magnitude)
    uint32_t k1[6], k2[6] = {0, 0, 0, 0, 0, 0}, t[6];
#pragma unroll
    for (int w = 0; w < 6; w++) k1[w] = k[w];
    zl_mul_words<4, 4, 6>(c1, a1, t);
    zl_addsub_words<6>(k1, t, P::SUB_C1A1 != 0);
    zl_mul_words<4, 4, 6>(c2, a2, t);
    zl_addsub_words<6>(k1, t, P::SUB_C2A2 != 0);
    zl_mul_words<4, 4, 6>(c1, b1, t);
    zl_addsub_words<6>(k2, t, P::SUB_C1B1 != 0);
    zl_mul_words<4, 4, 6>(c2, b2, t);
    zl_addsub_words<6>(k2, t, P::SUB_C2B2 != 0);
    uint32_t neg1 = k1[5] >> 31, neg2 = k2[5] >> 31;
    if (neg1) { uint32_t m[6] = {0, 0, 0, 0, 0, 0}; zl_addsub_words<6>(m, k1, true); for (int w = 0; w < 6; w++) k1[w] = m[w]; }
    if (neg2) { uint32_t m[6] = {0, 0, 0, 0, 0, 0}; zl_addsub_words<6>(m, k2, true); for (int w = 0; w < 6; w++) k2[w] = m[w]; }
    if (bad && (k1[4] | k1[5] | (k1[3] >> 31) | k2[4] | k2[5] | (k2[3] >> 31)) != 0u) atomicOr(bad, 2u);  // a half above 127 bits: cannot happen (see above); reported like a non-canonical scalar
    if ((k1[0] | k1[1] | k1[2] | k1[3]) == 0) neg1 = 0;
    if ((k2[0] | k2[1] | k2[2] | k2[3]) == 0) neg2 = 0;
    uint4* o1 = reinterpret_cast<uint4*>(out + (size_t)i * 8);
    uint4* o2 = reinterpret_cast<uint4*>(out + ((size_t)n + i) * 8);
    o1[0] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
    o1[1] = make_uint4(0, 0, 0, neg1 << 31);
    o2[0] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
    o2[1] = make_uint4(0, 0, 0, neg2 << 31);
}
// phib[i] = phi(P_i) = (beta x_i, y_i); the point at infinity (all-zero) stays itself
template <class G>
__global__ void __launch_bounds__(128) k_glv_phi(const Affine<typename G::F>* __restrict__ bases, uint32_t n, Affine<typename G::F>* __restrict__ phib) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = bases[i];
    if (!p.is_inf()) p.x = zl::canon(zl::mul(p.x, G::glv_beta()));
    phib[i] = p;
}

// ---- GLS for BLS12-381 G2: k = k0 + k1 |z| + k2 |z|^2 + k3 |z|^3 (digits below |z| < 2^64), k P = k0 P - k1 psi(P) + k2 psi^2(P) - k3 psi^3(P) ----------
// out: 4n records of 8 words -- record j n + i = digit j of scalar i in words 0..1, its sign (odd j: negative) in bit 31 of word 7.
// q = floor(x / |z|) for x < 2^256: Barrett with m = floor(2^320 / |z|) = 2^256 + mlow: ((x mlow >> 256) + x) >> 64, at most one short.
template <class P>
__device__ __forceinline__ uint64_t zl_divmod_z(uint32_t x[8]) {
    uint32_t pw[16];
    {
        uint64_t acc = 0;
        uint32_t top = 0;
#pragma unroll
        for (int col = 0; col < 15; col++) {
#pragma unroll
            for (int a = 0; a < 8; a++) {
                const int b = col - a;
                if (b < 0 || b > 7) continue;
                const uint64_t pr = (uint64_t)x[a] * P::barrett(b);
                acc += pr;
                top += acc < pr ? 1u : 0u;
            }
            pw[col] = (uint32_t)acc;
            acc = (acc >> 32) | ((uint64_t)top << 32);
            top = 0;
        }
        pw[15] = (uint32_t)acc;
    }
    uint32_t t[9];  // (x mlow >> 256) + x
    {
        uint64_t carry = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            carry += (uint64_t)pw[8 + w] + x[w];
            t[w] = (uint32_t)carry;
            carry >>= 32;
        }
        t[8] = (uint32_t)carry;
    }
    uint32_t q[8];
#pragma unroll
    for (int w = 0; w < 7; w++) q[w] = t[w + 2];
    q[7] = 0;
    // rem = x - q |z| (mod 2^128; the true value is below 2 |z| < 2^65)
    const uint64_t Z = (uint64_t)P::z(0) | ((uint64_t)P::z(1) << 32);
    const uint64_t q01 = (uint64_t)q[0] | ((uint64_t)q[1] << 32), q23 = (uint64_t)q[2] | ((uint64_t)q[3] << 32);
    const uint64_t lo = q01 * Z, hi = __umul64hi(q01, Z) + q23 * Z;
    const uint64_t x01 = (uint64_t)x[0] | ((uint64_t)x[1] << 32), x23 = (uint64_t)x[2] | ((uint64_t)x[3] << 32);
    uint64_t rlo = x01 - lo, rhi = x23 - hi - (x01 < lo ? 1u : 0u);
    if (rhi != 0 || rlo >= Z) {
        rhi -= rlo < Z ? 1u : 0u;
        rlo -= Z;
        uint32_t carry = 1;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            const uint32_t v = q[w] + carry;
            carry = (v < carry) ? 1u : 0u;
            q[w] = v;
        }
    }
#pragma unroll
    for (int w = 0; w < 8; w++) x[w] = q[w];
    return rlo;
}
template <class P>
__global__ void __launch_bounds__(256) k_gls_split(const uint32_t* __restrict__ scalars, uint32_t n, const uint8_t* __restrict__ inf, uint32_t* __restrict__ out,
                                                    int sc_bits, uint32_t* __restrict__ bad) {
    ZL_SIDE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
    uint4 lo4 = sp[0], hi4 = sp[1];
    zl_flag_wide_scalar(hi4.w, sc_bits, bad);
    if (inf && inf[i]) lo4 = hi4 = make_uint4(0, 0, 0, 0);
    uint32_t k[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    zl_reduce_once_mod_r<P>(k);  // k in [r, 2^255) may exceed |z|^4 - 1: the fourth quotient would not be a digit
    uint64_t d[4];
    d[0] = zl_divmod_z<P>(k);
    d[1] = zl_divmod_z<P>(k);
    d[2] = zl_divmod_z<P>(k);
    d[3] = (uint64_t)k[0] | ((uint64_t)k[1] << 32);  // k < r < |z|^4: the last quotient is a digit
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint4* o = reinterpret_cast<uint4*>(out + ((size_t)j * n + i) * 8);
        o[0] = make_uint4((uint32_t)d[j], (uint32_t)(d[j] >> 32), 0, 0);
        o[1] = make_uint4(0, 0, 0, (d[j] != 0 && (j & 1)) ? 0x80000000u : 0u);
    }
}
// phib[(j - 1) n + i] = psi^j(P_i), j = 1..3; psi(x, y) = (conj(x) gx, conj(y) gy); infinity (all-zero) stays itself
template <class G>
__global__ void __launch_bounds__(64) k_gls_psi(const Affine<typename G::F>* __restrict__ bases, uint32_t n, Affine<typename G::F>* __restrict__ phib) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = bases[i];
    const bool inf = p.is_inf();
    const F gx = G::psi_x(), gy = G::psi_y();
    for (int j = 0; j < 3; j++) {
        if (!inf) {
            F cx = p.x, cy = p.y;
            cx.c1 = zl::canon(zl::neg(cx.c1));  // conj: (c0, -c1); canonical again before it enters a product
            cy.c1 = zl::canon(zl::neg(cy.c1));
            p.x = zl::canon(zl::mul(cx, gx));
            p.y = zl::canon(zl::mul(cy, gy));
        }
        phib[(size_t)j * n + i] = p;
    }
}
