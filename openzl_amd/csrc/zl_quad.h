// zl_quad.h -- one group operation on the four lanes of a DPP quad (device only).
//
// The tails of an MSM (merge of cut buckets, bucket reduction levels) and every phase of a SMALL MSM are chains of dependent group
// operations with few independent ones beside them: the machine is empty and the time is the latency of one addition in one wave --
// 14.5 field products one after the other for a full XYZZ addition (a lone wave issues one v_mad_u64_u32 per ~6 cycles: 13-17 us per
// addition in G1, ~40 us in G2).  An addition's products are not a chain, though: its dependency depth is four.  Here the four lanes of a
// quad hold the same operands, each lane multiplies a DIFFERENT pair in the same instruction slot (operands picked by v_cndmask on the lane's
// position in the quad), and the four results are handed round with DPP quad_perm moves -- four product slots instead of 14.5 (full
// addition) or 10.5 (mixed addition).  Control flow (infinity, P = +-Q) depends on the operands only, i.e. is uniform inside a quad, which
// is all that DPP needs.  Work per addition goes up by ~25 % (4 lanes x 4.5 slots against 14.5), so this is for launches that do not fill the
// machine; the kernels take it as a template flag and the host picks it by lane count.
#pragma once
#include "zl_curve.h"

namespace zl {
template <int SRC>
__device__ __forceinline__ uint32_t quad_bcast_u32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, SRC * 0x55, 0xf, 0xf, true);  // quad_perm:[SRC,SRC,SRC,SRC]
#else
    return v;
#endif
}
__device__ __forceinline__ uint32_t quad_sel_u32(int sub, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t lo = (sub & 1) ? x1 : x0, hi = (sub & 1) ? x3 : x2;
    return (sub & 2) ? hi : lo;
}
// value of lane SRC of the quad, in every lane of the quad
template <int SRC, class A, class B>
__device__ __forceinline__ Fp28<A, B> quad_bcast(const Fp28<A, B>& v) {
    Fp28<A, B> r = v;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = quad_bcast_u32<SRC>(v.l[i]);
    return r;
}
template <int SRC, class P>
__device__ __forceinline__ Fp<P> quad_bcast(const Fp<P>& v) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = quad_bcast_u32<SRC>(v.l[i]);
    return r;
}
template <int SRC, class B, bool I>
__device__ __forceinline__ Fp2LT<B, I> quad_bcast(const Fp2LT<B, I>& v) {
    return Fp2LT<B, I>{quad_bcast<SRC>(v.c0), quad_bcast<SRC>(v.c1)};
}
template <int SRC, class P>
__device__ __forceinline__ Fp2<P> quad_bcast(const Fp2<P>& v) {
    return Fp2<P>{quad_bcast<SRC>(v.c0), quad_bcast<SRC>(v.c1)};
}
// operand of this lane: x_sub
template <class A, class B>
__device__ __forceinline__ Fp28<A, B> quad_sel(int sub, const Fp28<A, B>& x0, const Fp28<A, B>& x1, const Fp28<A, B>& x2, const Fp28<A, B>& x3) {
    Fp28<A, B> r = x0;
#pragma unroll
    for (int i = 0; i < A::L; i++) r.l[i] = quad_sel_u32(sub, x0.l[i], x1.l[i], x2.l[i], x3.l[i]);
    return r;
}
template <class P>
__device__ __forceinline__ Fp<P> quad_sel(int sub, const Fp<P>& x0, const Fp<P>& x1, const Fp<P>& x2, const Fp<P>& x3) {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.l[i] = quad_sel_u32(sub, x0.l[i], x1.l[i], x2.l[i], x3.l[i]);
    return r;
}
template <class B, bool I>
__device__ __forceinline__ Fp2LT<B, I> quad_sel(int sub, const Fp2LT<B, I>& x0, const Fp2LT<B, I>& x1, const Fp2LT<B, I>& x2, const Fp2LT<B, I>& x3) {
    return Fp2LT<B, I>{quad_sel(sub, x0.c0, x1.c0, x2.c0, x3.c0), quad_sel(sub, x0.c1, x1.c1, x2.c1, x3.c1)};
}
template <class P>
__device__ __forceinline__ Fp2<P> quad_sel(int sub, const Fp2<P>& x0, const Fp2<P>& x1, const Fp2<P>& x2, const Fp2<P>& x3) {
    return Fp2<P>{quad_sel(sub, x0.c0, x1.c0, x2.c0, x3.c0), quad_sel(sub, x0.c1, x1.c1, x2.c1, x3.c1)};
}

// p += q on a quad (add-2008-s, the operation sequence and the bounds of add_full in zl_curve.h; every lane of the quad passes the same p, q and
// ends with the same p).  sub = position of the lane in its quad.
template <class F>
__device__ __forceinline__ void add_full_quad(XYZZ<F>& p, const XYZZ<F>& q, int sub) {
    if (q.is_inf()) return;
    if (p.is_inf()) { p = q; return; }
    // slot 1: u1 = x1 zz2 | u2 = x2 zz1 | s1 = y1 zzz2 | s2 = y2 zzz1                      64 -> < 2
    F t = mul(quad_sel(sub, p.x, q.x, p.y, q.y), quad_sel(sub, q.zz, p.zz, q.zzz, p.zzz));
    const F u1 = quad_bcast<0>(t), u2 = quad_bcast<1>(t), s1 = quad_bcast<2>(t), s2 = quad_bcast<3>(t);
    const F pp_ = subk<1>(u2, u1), r = subk<1>(s2, s1);                                     // < 4
    if (pp_.is_zero()) {
        if (r.is_zero()) { dbl_inplace(p); return; }
        p = XYZZ<F>::inf();
        return;
    }
    // slot 2: pp = pp_^2 | rr = r^2 | zz1 zz2 | zzz1 zzz2                                    16, 16, 64, 64 -> < 2
    t = mul(quad_sel(sub, pp_, r, p.zz, p.zzz), quad_sel(sub, pp_, r, q.zz, q.zzz));
    const F pp = quad_bcast<0>(t), rr = quad_bcast<1>(t), zz12 = quad_bcast<2>(t), zzz12 = quad_bcast<3>(t);
    // slot 3: ppp = pp_ pp | q = u1 pp | zz3 = zz12 pp                                        8, 4, 4 -> < 2
    t = mul(quad_sel(sub, pp_, u1, zz12, zz12), pp);
    const F ppp = quad_bcast<0>(t), q_ = quad_bcast<1>(t), zz3 = quad_bcast<2>(t);
    constexpr int J = ScanBias<F>::J;
    const F x3 = x3_of(rr, ppp, q_);                                                        // < 8
    // slot 4: y3 = r (q - x3) - s1 ppp | zzz3 = zzz12 ppp (+ 0 ppp)                          4*10 + 4*2 [4*18 + 4*2] -> < 2
    const F zero = F::zero();
    t = muladd(quad_sel(sub, r, zzz12, r, zzz12), quad_sel(sub, subk_scan<J>(q_, x3), ppp, ppp, ppp), quad_sel(sub, negk_scan<2>(s1), zero, zero, zero), ppp);
    p.y = quad_bcast<0>(t);
    p.zzz = quad_bcast<1>(t);
    p.x = x3;
    p.zz = zz3;
}
// p += (qx, qy) on a quad (madd-2008-s, sequence and bounds of add_mixed)
template <class F>
__device__ __forceinline__ void add_mixed_quad(XYZZ<F>& p, const F& qx, const F& qy_in, bool neg_q, int sub) {
    constexpr int J = ScanBias<F>::J;
    if (p.is_inf()) {
        p.x = qx; p.y = neg_q ? negk<1>(qy_in) : qy_in; p.zz = F::one(); p.zzz = F::one();
        return;
    }
    const F qy = neg_q ? negk_scan<2>(qy_in) : qy_in;                                       // < 4
    // slot 1: u2 = x2 zz1 | s2 = y2 zzz1                                                      16, 32 -> < 2
    F t = mul(quad_sel(sub, qx, qy, qx, qy), quad_sel(sub, p.zz, p.zzz, p.zz, p.zzz));
    const F u2 = quad_bcast<0>(t), s2 = quad_bcast<1>(t);
    const F pp_ = subk<3>(u2, p.x), r = subk<3>(s2, p.y);                                   // < 10
    if (pp_.is_zero()) {
        if (r.is_zero()) { p = dbl_affine(qx, neg_q ? negk<1>(qy_in) : qy_in); return; }
        p = XYZZ<F>::inf();
        return;
    }
    // slot 2: pp = pp_^2 | rr = r^2                                                           100 -> < 2
    t = mul(quad_sel(sub, pp_, r, pp_, r), quad_sel(sub, pp_, r, pp_, r));
    const F pp = quad_bcast<0>(t), rr = quad_bcast<1>(t);
    // slot 3: ppp = pp_ pp | q = x1 pp | zz3 = zz1 pp                                          20, 16, 16 -> < 2
    t = mul(quad_sel(sub, pp_, p.x, p.zz, p.zz), pp);
    const F ppp = quad_bcast<0>(t), q = quad_bcast<1>(t), zz3 = quad_bcast<2>(t);
    const F x3 = x3_of(rr, ppp, q);                                                         // < 8
    // slot 4: y3 = r (q - x3) - y1 ppp | zzz3 = zzz1 ppp                                       10*10 + 8*2 [10*18 + 16*2] -> < 2
    const F zero = F::zero();
    t = muladd(quad_sel(sub, r, p.zzz, r, p.zzz), quad_sel(sub, subk_scan<J>(q, x3), ppp, ppp, ppp), quad_sel(sub, negk_scan<J>(p.y), zero, zero, zero), ppp);
    p.y = quad_bcast<0>(t);
    p.zzz = quad_bcast<1>(t);
    p.x = x3;
    p.zz = zz3;
}
}  // namespace zl
