// zl_field.h -- Montgomery prime-field arithmetic on 32-bit limbs for gfx950 (and the host tail code).
//
// Replaces, for this backend, what ark-ff 0.3.0 Fp256/Fp384 does on the CPU (`pub use ff::*`,
// /root/reference/plugins/arkworks/src/ff.rs:6): same value*R mod p representation with
// R = 2^(64*ceil(bits/64)), so limbs are byte-compatible with arkworks' in-memory BigInteger limbs
// (two u32 = one little-endian u64).  Everything is fully reduced to [0,p) on output.
//
// gfx950 has no 64-bit multiplier; the work-horse is v_mad_u64_u32 (32x32+64 -> 64).  The CIOS loops below
// are written so that each inner step is exactly one such mad plus a 64-bit carry add.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZL_HD __host__ __device__ __forceinline__
#else
#define ZL_HD inline
#endif
#include "zl_params.h"
#include "zl_mul_gfx950.h"

template <class P>
struct alignas(16) Fp {
    static constexpr int N = P::N;
    uint32_t l[N];

    ZL_HD static Fp zero() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    ZL_HD static Fp one() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::one(i);
        return r;
    }
    ZL_HD static Fp r2() {
        Fp r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::r2(i);
        return r;
    }
    ZL_HD bool is_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i];
        return acc == 0;
    }
    ZL_HD bool operator==(const Fp& o) const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < N; i++) acc |= l[i] ^ o.l[i];
        return acc == 0;
    }
    ZL_HD bool operator!=(const Fp& o) const { return !(*this == o); }
    ZL_HD bool raw_zero() const { return is_zero(); }  // limbs all zero (fully reduced representation)
};

namespace zl {

// r = a - p if a >= p else a   (a < 2p)
template <class P>
ZL_HD void reduce_once(uint32_t* a) {
    constexpr int N = P::N;
    uint32_t t[N];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a[i] - P::mod(i) - br;
        t[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    // br == 1 -> a < p -> keep a
#pragma unroll
    for (int i = 0; i < N; i++) a[i] = br ? a[i] : t[i];
}

template <class P>
ZL_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    reduce_once<P>(r.l);  // every modulus here leaves >= 1 spare top bit: no carry out
    return r;
}
template <class P>
ZL_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    Fp<P> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
        r.l[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)br;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        c += (uint64_t)r.l[i] + (P::mod(i) & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
template <class P>
ZL_HD Fp<P> dbl(const Fp<P>& a) {
    return add(a, a);
}
// bias-annotated spellings shared with the lazily reduced 28-bit field (zl_field28.h): no-ops here, every value is < p
template <int J, class P>
ZL_HD Fp<P> subk(const Fp<P>& a, const Fp<P>& b) {
    return sub(a, b);
}
template <class P>
ZL_HD Fp<P> wred(const Fp<P>& a) {
    return a;
}
template <class P>
ZL_HD Fp<P> canon(const Fp<P>& a) {
    return a;
}
template <class P>
ZL_HD Fp<P> neg(const Fp<P>& a) {
    return sub(Fp<P>::zero(), a);  // a == 0: no borrow, stays 0; otherwise p - a
}
template <int J, class P>
ZL_HD Fp<P> negk(const Fp<P>& a) {
    return neg(a);
}

// Montgomery product a*b*R^-1 mod p, CIOS on 32-bit limbs.  All moduli have their top bit clear, so the
// running value stays < 2p < 2^(32N) and no extra carry word is needed.
template <class P>
ZL_HD Fp<P> mul_body(const Fp<P>& a, const Fp<P>& b) {
    constexpr int N = P::N;
    uint32_t t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        // invariant: T = sum t[j] 2^(32j) < 2p < 2^(32N)  (T_i = (a*(b mod 2^(32i)) + M_i*p) / 2^(32i), M_i < 2^(32i))
        uint64_t c = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            uint64_t p = (uint64_t)a.l[j] * bi + t[j] + c;
            t[j] = (uint32_t)p;
            c = p >> 32;
        }
        const uint32_t tn = (uint32_t)c;  // T + a*b_i < 2p + p*2^32: top word fits 32 bits
        const uint32_t m = t[0] * P::INV;
        uint64_t p = (uint64_t)m * P::mod(0) + t[0];
        c = p >> 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            p = (uint64_t)m * P::mod(j) + t[j] + c;
            t[j - 1] = (uint32_t)p;
            c = p >> 32;
        }
        t[N - 1] = tn + (uint32_t)c;  // new T < 2p: no carry out
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = t[i];
    reduce_once<P>(r.l);
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 multiplier: product-scanning (column-wise) Montgomery with a 96-bit column accumulator -------------
// One MAC = v_mad_u64_u32 (64-bit accumulate, carry-out to VCC) + v_addc_co_u32 into the third word: no
// zero-extension moves, no 64-bit adds.  Fully expanded by gen_mul.py (zl_mul_gfx950.h).
template <class P>
__device__ __forceinline__ Fp<P> mul_dev(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> out;
    if constexpr (P::N == 8) mul_dev_8<P>(out.l, a.l, b.l);
    else mul_dev_12<P>(out.l, a.l, b.l);
    reduce_once<P>(out.l);
    return out;
}
#endif

// The multiplier is ONE out-of-line function per field (arguments and result travel in VGPRs on the device):
// a fully inlined point addition is ~50 KB of code per call site, which overflows the instruction cache and
// takes hipcc tens of minutes to schedule; a called 5 KB body stays I-cache resident.
#if defined(__HIPCC__)
#define ZL_NOINLINE_HD __host__ __device__ __attribute__((noinline))
#else
#define ZL_NOINLINE_HD __attribute__((noinline))
#endif
// Operands are passed as scalar u32 arguments so that the AMDGPU calling convention keeps all of them in VGPRs
// (v0..v23): passing two 48-byte structs by value sent the second one through scratch memory (3 x 16-B stores + loads
// + an s_waitcnt vmcnt(0) per call; 14.8 GB of scratch writes per 2^24 MSM in the first PMC pass).
#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host: the same Montgomery product (same R = 2^(32N), same fully reduced result) on N/2 limbs of 64 bits with 128-bit accumulators:
// the host tails (window Horner, Groth16's blinding terms, normalisations) and the witness arithmetic run 3-4x faster than on the 32-bit scan.
template <class P>
inline Fp<P> mul_host64(const Fp<P>& a, const Fp<P>& b) {
    typedef unsigned __int128 u128;
    constexpr int M = P::N / 2;
    static_assert(P::N % 2 == 0, "even limb count");
    uint64_t x[M], y[M], q[M], t[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
        x[j] = (uint64_t)a.l[2 * j] | ((uint64_t)a.l[2 * j + 1] << 32);
        y[j] = (uint64_t)b.l[2 * j] | ((uint64_t)b.l[2 * j + 1] << 32);
        q[j] = (uint64_t)P::mod(2 * j) | ((uint64_t)P::mod(2 * j + 1) << 32);
        t[j] = 0;
    }
    uint64_t pinv = (uint64_t)(0u - P::INV);  // p^-1 mod 2^32, one Newton step to 2^64
    pinv *= 2 - q[0] * pinv;
    const uint64_t ninv = 0 - pinv;
#pragma unroll
    for (int i = 0; i < M; i++) {
        u128 c = 0;
#pragma unroll
        for (int j = 0; j < M; j++) {
            c += (u128)x[j] * y[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        const uint64_t tn = (uint64_t)c;
        const uint64_t m = t[0] * ninv;
        c = ((u128)m * q[0] + t[0]) >> 64;
#pragma unroll
        for (int j = 1; j < M; j++) {
            c += (u128)m * q[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        t[M - 1] = tn + (uint64_t)c;  // < 2p: no carry out (top bit of every modulus is clear)
    }
    Fp<P> r;
#pragma unroll
    for (int j = 0; j < M; j++) {
        r.l[2 * j] = (uint32_t)t[j];
        r.l[2 * j + 1] = (uint32_t)(t[j] >> 32);
    }
    reduce_once<P>(r.l);
    return r;
}
#endif
template <class P>
ZL_HD Fp<P> mul_impl(const Fp<P>& a, const Fp<P>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_PORTABLE_MUL)
    return mul_dev(a, b);
#elif !defined(__HIP_DEVICE_COMPILE__) && !defined(ZL_PORTABLE_MUL)
    return mul_host64(a, b);
#else
    return mul_body(a, b);
#endif
}
template <class P>
ZL_NOINLINE_HD Fp<P> mul_call8(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7) {
    Fp<P> a, b;
    a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7;
    b.l[0] = b0; b.l[1] = b1; b.l[2] = b2; b.l[3] = b3; b.l[4] = b4; b.l[5] = b5; b.l[6] = b6; b.l[7] = b7;
    return mul_impl(a, b);
}
template <class P>
ZL_NOINLINE_HD Fp<P> sqr_call8(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
    Fp<P> a;
    a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7;
    return mul_impl(a, a);
}
template <class P>
ZL_NOINLINE_HD Fp<P> mul_call12(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t b4, uint32_t b5, uint32_t b6, uint32_t b7, uint32_t b8, uint32_t b9, uint32_t b10, uint32_t b11) {
    Fp<P> a, b;
    a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7; a.l[8] = a8; a.l[9] = a9; a.l[10] = a10; a.l[11] = a11;
    b.l[0] = b0; b.l[1] = b1; b.l[2] = b2; b.l[3] = b3; b.l[4] = b4; b.l[5] = b5; b.l[6] = b6; b.l[7] = b7; b.l[8] = b8; b.l[9] = b9; b.l[10] = b10; b.l[11] = b11;
    return mul_impl(a, b);
}
template <class P>
ZL_NOINLINE_HD Fp<P> sqr_call12(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7, uint32_t a8, uint32_t a9, uint32_t a10, uint32_t a11) {
    Fp<P> a;
    a.l[0] = a0; a.l[1] = a1; a.l[2] = a2; a.l[3] = a3; a.l[4] = a4; a.l[5] = a5; a.l[6] = a6; a.l[7] = a7; a.l[8] = a8; a.l[9] = a9; a.l[10] = a10; a.l[11] = a11;
    return mul_impl(a, a);
}
template <class P>
ZL_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return mul_impl(a, b);  // host: the 64-bit product inline (the out-of-line entry with its 16 scalar arguments exists for the device's register allocation)
#elif !defined(ZL_INLINE_MUL) && !(defined(ZL_INLINE_MUL_DEVICE) && defined(__HIP_DEVICE_COMPILE__))
    if constexpr (P::N == 8) return mul_call8<P>(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7]);
    else return mul_call12<P>(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11], b.l[0], b.l[1], b.l[2], b.l[3], b.l[4], b.l[5], b.l[6], b.l[7], b.l[8], b.l[9], b.l[10], b.l[11]);
#else
    return mul_impl(a, b);
#endif
}
template <class P>
ZL_HD Fp<P> sqr(const Fp<P>& a) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return mul_impl(a, a);
#elif !defined(ZL_INLINE_MUL) && !(defined(ZL_INLINE_MUL_DEVICE) && defined(__HIP_DEVICE_COMPILE__))
    if constexpr (P::N == 8) return sqr_call8<P>(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7]);
    else return sqr_call12<P>(a.l[0], a.l[1], a.l[2], a.l[3], a.l[4], a.l[5], a.l[6], a.l[7], a.l[8], a.l[9], a.l[10], a.l[11]);
#else
    return mul_impl(a, a);
#endif
}

template <class P>
ZL_HD Fp<P> to_mont(const Fp<P>& canon) {
    return mul(canon, Fp<P>::r2());
}
template <class P>
ZL_HD Fp<P> from_mont(const Fp<P>& a) {
    Fp<P> o = Fp<P>::zero();
    o.l[0] = 1;
    return mul(a, o);
}
// a^e for a little-endian exponent of `nbits` bits held in 32-bit words
template <class P>
ZL_HD Fp<P> pow_words(const Fp<P>& a, const uint32_t* e, int nbits) {
    Fp<P> acc = Fp<P>::one();
    for (int i = nbits - 1; i >= 0; i--) {
        acc = sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, a);
    }
    return acc;
}
// Fermat inverse a^(p-2); inv(0) = 0
template <class P>
ZL_HD Fp<P> inv(const Fp<P>& a) {
    constexpr int N = P::N;
    uint32_t e[N];
    uint32_t borrow = 2;  // e = p - 2 with borrow propagation (the low limb of BLS12-381 Fr is 1)
    for (int i = 0; i < N; i++) {
        const uint32_t m = P::mod(i);
        e[i] = m - borrow;
        borrow = m < borrow ? 1u : 0u;
    }
    return pow_words(a, e, 32 * N);
}
template <class P>
ZL_HD Fp<P> from_u64(uint64_t v) {
    Fp<P> c = Fp<P>::zero();
    c.l[0] = (uint32_t)v;
    c.l[1] = (uint32_t)(v >> 32);
    return to_mont(c);
}

}  // namespace zl
