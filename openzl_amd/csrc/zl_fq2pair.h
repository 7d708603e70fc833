// zl_fq2pair.h -- Fq2 = Fq[u]/(u^2 + 1) split over a PAIR of lanes (device only; round 6).
//
// The G2 kernels of rounds 2-5 keep both components of every Fq2 coordinate in one lane: the bucket accumulation then holds 416 registers, i.e. ONE
// wave per SIMD (a lone wave issues one instruction per 5.2-5.9 cycles where three waves reach one per 3.2), and its 10.6 k mads per mixed addition are
// an 85-KB instruction stream against a 64-KB instruction cache; the tails spill 540-780 B per lane.  Here lane i of a row of 16 holds the c0 component
// and lane i ^ 8 the c1 component of the SAME element ("half" 0 / 1): registers per lane and the instruction stream of an addition halve, the mads
// per Fq2 product stay what they were (each lane runs ONE dual product scan: c0 = a0 b0 - a1 b1 | c1 = a1 b0 + a0 b1), and the operands of the other
// half arrive by DPP row_ror:8 -- a rotation by eight inside a row of 16 is exactly the swap i <-> i ^ 8, and the DPP bank mask (banks of four lanes:
// 0x3 = lanes 0-7 = half 0, 0xC = lanes 8-15 = half 1) makes the move conditional on the half for free, so the operand routing of a product costs
// 14 subtractions + 42 DPP moves beside 588 mads and no v_cndmask at all.
//
// Fp2H<B> has the contracts of Fp2LT<B, .> component-wise (zl_curve.h): mul / sqr / muladd take components <= 16q and return components < 2q, so the
// point formulas of zl_curve.h (add_mixed, add_full, dbl_*) apply unchanged and with the bounds proved in zl_bounds.h.  Every predicate (is_zero,
// raw_zero) is combined over the pair, so control flow stays uniform inside a pair -- which is all DPP needs.
#pragma once
#include "zl_curve.h"
#include "zl_quad.h"

namespace zl {
// (the host pass of a HIP unit parses these bodies too: it sees plain moves)
#if defined(__HIP_DEVICE_COMPILE__)
#define ZL_PAIR_DPP(keep, v, bank) (uint32_t) __builtin_amdgcn_update_dpp((int)(keep), (int)(v), 0x128, 0xf, bank, false)  // row_ror:8
#else
#define ZL_PAIR_DPP(keep, v, bank) (keep)
#endif
// the partner lane's v
__device__ __forceinline__ uint32_t pair_other_u32(uint32_t v) { return ZL_PAIR_DPP(v, v, 0xf); }
// half 1 takes the partner's v, half 0 keeps `keep` (and the other way round)
__device__ __forceinline__ uint32_t pair_take_hi_u32(uint32_t keep, uint32_t v) { return ZL_PAIR_DPP(keep, v, 0xc); }
__device__ __forceinline__ uint32_t pair_take_lo_u32(uint32_t keep, uint32_t v) { return ZL_PAIR_DPP(keep, v, 0x3); }
__device__ __forceinline__ int pair_half() { return (int)((threadIdx.x >> 3) & 1u); }  // one-dimensional blocks whose size is a multiple of 16
// (in namespace zl: the point formulas of zl_curve.h find the operations below by argument-dependent lookup at instantiation)
template <class B>
struct Fp2H {
    B c;  // c0 of the element in the lanes of half 0, c1 in the lanes of half 1
    __device__ __forceinline__ static Fp2H zero() { return Fp2H{B::zero()}; }
    __device__ __forceinline__ static Fp2H one() {
        B o = B::one();
        const bool hi = pair_half() != 0;
#pragma unroll
        for (int i = 0; i < B::L; i++) o.l[i] = hi ? 0u : o.l[i];
        return Fp2H{o};
    }
    // all limbs of both components zero (stored canonical zero / the infinity encodings)
    __device__ __forceinline__ bool raw_zero() const {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < B::L; i++) acc |= c.l[i];
        acc |= pair_other_u32(acc);
        return acc == 0;
    }
    __device__ __forceinline__ bool is_zero() const {  // == 0 mod q in both components
        const uint32_t z = c.is_zero() ? 1u : 0u;
        return (z & pair_other_u32(z)) != 0;
    }
};
template <class B> __device__ __forceinline__ Fp2H<B> add(const Fp2H<B>& a, const Fp2H<B>& b) { return Fp2H<B>{add(a.c, b.c)}; }
template <class B> __device__ __forceinline__ Fp2H<B> dbl(const Fp2H<B>& a) { return Fp2H<B>{dbl(a.c)}; }
template <int J, class B> __device__ __forceinline__ Fp2H<B> subk(const Fp2H<B>& a, const Fp2H<B>& b) { return Fp2H<B>{subk<J>(a.c, b.c)}; }
template <int J, class B> __device__ __forceinline__ Fp2H<B> negk(const Fp2H<B>& a) { return Fp2H<B>{negk<J>(a.c)}; }
template <class B> __device__ __forceinline__ Fp2H<B> sub(const Fp2H<B>& a, const Fp2H<B>& b) { return subk<4>(a, b); }
template <class B> __device__ __forceinline__ Fp2H<B> neg(const Fp2H<B>& a) { return negk<4>(a); }
template <class B> __device__ __forceinline__ Fp2H<B> wred(const Fp2H<B>& a) { return Fp2H<B>{wred(a.c)}; }
template <class B> __device__ __forceinline__ Fp2H<B> canon(const Fp2H<B>& a) { return Fp2H<B>{canon(a.c)}; }
template <class B> __device__ __forceinline__ Fp2H<B> x3_of(const Fp2H<B>& rr, const Fp2H<B>& ppp, const Fp2H<B>& q) { return Fp2H<B>{x3_of(rr.c, ppp.c, q.c)}; }

// the routing of one right-hand operand b: S = b0 in both halves, T = -b1 (32q - b1, un-carried: scan-only) in half 0 and b1 in half 1,
// so that own_a * S + other_a * T is a0 b0 - a1 b1 in half 0 and a1 b0 + a0 b1 in half 1.  b's components carried and <= 16q.
template <class B>
__device__ __forceinline__ void pair_route(const B& b, B& S, B& T) {
    const B nb = negk_scan<5>(b);
    S = b;
    T = b;
#pragma unroll
    for (int i = 0; i < B::L; i++) {
        S.l[i] = pair_take_hi_u32(b.l[i], b.l[i]);
        T.l[i] = pair_take_lo_u32(b.l[i], nb.l[i]);
    }
}
template <class B>
__device__ __forceinline__ B pair_other(const B& a) {
    B o = a;
#pragma unroll
    for (int i = 0; i < B::L; i++) o.l[i] = pair_other_u32(a.l[i]);
    return o;
}
// (a0 + a1 u)(b0 + b1 u): ONE dual product scan per lane.  16*16 + 16*32 <= 2500 -> components < 2q
template <class B>
__device__ __forceinline__ Fp2H<B> mul(const Fp2H<B>& a, const Fp2H<B>& b) {
    B S, T;
    pair_route(b.c, S, T);
    return Fp2H<B>{muladd(a.c, S, pair_other(a.c), T)};
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u: ONE product scan per lane, X * Y with
//   X = a0 + other   (half 0: a0 + a1; half 1: 2 a0)          un-carried, < 32q
//   Y = half 0: own - other + 32q (un-carried); half 1: own   < 48q
// a column of 14 products of limbs < 2^29 by limbs < 2^30 plus 14 of the reduction stays below 2^63.  32 * 48 <= 2500 -> < 2q
template <class B>
__device__ __forceinline__ Fp2H<B> sqr(const Fp2H<B>& a) {
    const B o = pair_other(a.c);
    const B d = subk_scan<5>(a.c, o);
    const bool hi = pair_half() != 0;
    B X = a.c, Y = a.c;
#pragma unroll
    for (int i = 0; i < B::L; i++) {
        X.l[i] = pair_take_hi_u32(a.c.l[i], a.c.l[i]) + o.l[i];
        Y.l[i] = hi ? a.c.l[i] : d.l[i];
    }
    return Fp2H<B>{mul(X, Y)};
}
// a b + c d with one reduction per component: ONE four-product scan per lane.  2 * (16*16 + 16*32) <= 2500 -> < 2q
template <class B>
__device__ __forceinline__ Fp2H<B> muladd(const Fp2H<B>& a, const Fp2H<B>& b, const Fp2H<B>& c, const Fp2H<B>& d) {
    B Sb, Tb, Sd, Td;
    pair_route(b.c, Sb, Tb);
    pair_route(d.c, Sd, Td);
    return Fp2H<B>{muladd4(a.c, Sb, pair_other(a.c), Tb, c.c, Sd, pair_other(c.c), Td)};
}
// four lanes per group operation (zl_quad.h) ON TOP of the pair split: the quads {4k .. 4k+3} and {4k+8 .. 4k+11} of a row hold the two halves of one
// element, quad_perm moves stay inside a quad and row_ror:8 maps quad k onto quad k + 2 lane for lane -- eight lanes per Fq2 group operation
template <int SRC, class B>
__device__ __forceinline__ Fp2H<B> quad_bcast(const Fp2H<B>& v) { return Fp2H<B>{quad_bcast<SRC>(v.c)}; }
template <class B>
__device__ __forceinline__ Fp2H<B> quad_sel(int sub, const Fp2H<B>& x0, const Fp2H<B>& x1, const Fp2H<B>& x2, const Fp2H<B>& x3) {
    return Fp2H<B>{quad_sel(sub, x0.c, x1.c, x2.c, x3.c)};
}
}  // namespace zl

using zl::Fp2H;
template <class B> struct HotField<Fp2H<B>> { using type = Fp2H<B>; };

// ---- memory: an Fq2 element is {c0, c1}, each one B (64 bytes); a lane moves the component of its half ---------------------------------
template <class B>
__device__ __forceinline__ Fp2H<B> pair_load(const Fp2LT<B, false>* p, int half) {
    return Fp2H<B>{reinterpret_cast<const B*>(p)[half]};
}
template <class B>
__device__ __forceinline__ void pair_store(Fp2LT<B, false>* p, int half, const Fp2H<B>& v) {
    reinterpret_cast<B*>(p)[half] = v.c;
}
template <class B>
__device__ __forceinline__ XYZZ<Fp2H<B>> pair_load(const XYZZ<Fp2LT<B, false>>* p, int half) {
    return XYZZ<Fp2H<B>>{pair_load(&p->x, half), pair_load(&p->y, half), pair_load(&p->zz, half), pair_load(&p->zzz, half)};
}
template <class B>
__device__ __forceinline__ void pair_store(XYZZ<Fp2LT<B, false>>* p, int half, const XYZZ<Fp2H<B>>& v) {
    pair_store(&p->x, half, v.x);
    pair_store(&p->y, half, v.y);
    pair_store(&p->zz, half, v.zz);
    pair_store(&p->zzz, half, v.zzz);
}
template <class B>
__device__ __forceinline__ Affine<Fp2H<B>> pair_load(const Affine<Fp2LT<B, false>>* p, int half) {
    return Affine<Fp2H<B>>{pair_load(&p->x, half), pair_load(&p->y, half)};
}
// the component type of an Fq2 field on 28-bit limbs (void for every other field: the pair kernels exist for those groups only)
template <class F> struct PairBase { using type = void; };
template <class B, bool I> struct PairBase<Fp2LT<B, I>> { using type = B; };

// item / position of a lane in the pair kernels: 32 items per wave (PAIR: lanes i, i ^ 8), or 8 items per wave with four lanes per half (OCTET: a quad and the quad eight lanes on)
#define ZL_PAIR_ITEM() (blockIdx.x * 32u + ((threadIdx.x >> 4) << 3) + (threadIdx.x & 7u))
#define ZL_OCTET_ITEM() (blockIdx.x * 8u + ((threadIdx.x >> 4) << 1) + ((threadIdx.x >> 2) & 1u))
