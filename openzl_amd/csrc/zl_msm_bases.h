// zl_msm_bases.h -- kernels behind the bases handles: import / export of affine points, infinity flags, fixed-base generation k_i G,
// batch normalisation to affine coordinates and the window tables 2^(c w) P_i (zl_bases_precompute).  Instantiated in zl_msm.hip.
#pragma once
#include "zl_ctx.h"
#include "zl_msm_common.h"

// ------------------------------------------------------------------------------------------------ bases kernels
// canonical / Montgomery host records -> device Affine<F> (Montgomery).  in: packed x||y u32 limbs per point.
template <class G>
__global__ void __launch_bounds__(128) k_bases_import(const uint32_t* __restrict__ in, const uint8_t* __restrict__ inf_flags, uint32_t n, int to_mont,
                                                       int check, Affine<typename G::F>* __restrict__ out, uint32_t* __restrict__ bad) {
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int WORDS = FieldIO<F>::WORDS;  // 32-bit words per coordinate in the ABI layout
    const uint32_t* src = in + (size_t)i * 2 * WORDS;
    uint32_t acc = 0;
    for (int k = 0; k < 2 * WORDS; k++) acc |= src[k];
    const bool inf = acc == 0 || (inf_flags && inf_flags[i]);
    if (inf) { out[i] = Affine<F>::inf(); return; }
    Affine<F> p;
    if (to_mont) { p.x = FieldIO<F>::load_canon(src); p.y = FieldIO<F>::load_canon(src + WORDS); }
    else { p.x = FieldIO<F>::load_mont32(src); p.y = FieldIO<F>::load_mont32(src + WORDS); }
    if (check) {
        F lhs = zl::sqr(p.y);
        F rhs = zl::add(zl::mul(zl::sqr(p.x), p.x), G::coeff_b());
        if (lhs != rhs) atomicAdd(bad, 1u);
    }
    out[i] = p;
}
// flags[i] = 1 if bases[i] is the point at infinity; *count = how many (Groth16 query vectors hold many: a variable that appears in no row
// of B has b_query[i] = 0 * G).  The recoder drops their scalars, so they cost neither a sort entry nor a (divergent, idle) accumulation step.
template <class G>
__global__ void __launch_bounds__(256) k_bases_inf_flags(const Affine<typename G::F>* __restrict__ in, uint32_t n, uint8_t* __restrict__ flags, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inf = i < n && in[i].is_inf();
    if (i < n) flags[i] = inf ? 1 : 0;
    const uint64_t m = __ballot(inf);
    if (m && (threadIdx.x & 63u) == 0) atomicAdd(count, (uint32_t)__popcll(m));
}
// out[i] = canonical affine x||y of bases[i]
template <class G>
__global__ void __launch_bounds__(128) k_bases_export(const Affine<typename G::F>* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int WORDS = FieldIO<F>::WORDS;
    const Affine<F> p = in[i];
    uint32_t* dst = out + (size_t)i * 2 * WORDS;
    if (p.is_inf()) { for (int k = 0; k < 2 * WORDS; k++) dst[k] = 0; return; }
    FieldIO<F>::store_canon(dst, p.x);
    FieldIO<F>::store_canon(dst + WORDS, p.y);
}
// ---- fixed-base windowed generation + batch normalisation (the setup path: ark_ec::msm::FixedBaseMSM::{get_window_table,
// multi_scalar_mul} + ProjectiveCurve::batch_normalization_into_affine behind Groth16::compile,
// /root/reference/plugins/arkworks/src/groth16.rs:427-443) -------------------------------------------------------
// T[w][d] = d * 2^(ZL_FB_BITS w) * G (affine), shared by every point: k * G is then ceil(256 / ZL_FB_BITS) mixed additions of table
// entries instead of 256 doublings + ~128 additions, and the affine conversion shares ONE field inversion among the ~32-64 points a
// lane normalises (Montgomery's trick) instead of one 570-multiplication Fermat inversion per point.
#ifndef ZL_FB_BITS
#define ZL_FB_BITS 8
#endif
#define ZL_FB_WINDOWS ((256 + ZL_FB_BITS - 1) / ZL_FB_BITS)
// Out-of-line group operations for the cold table-construction kernels: with Fq2 coordinates a fully inlined doubling + addition loop
// needs the whole 512-register budget plus spills, and hipcc (ROCm 7.2) was observed to drop one limb of a spilled coordinate in that
// shape (k_fb_table<BlsG2>: zz.c0.l[11] written as 0).  One call per operation keeps the kernels small; they are not on any timed path.
template <class F> __device__ __noinline__ void zl_add_full_ool(XYZZ<F>* p, const XYZZ<F>* q) { zl::add_full(*p, *q); }
template <class F> __device__ __noinline__ void zl_dbl_ool(XYZZ<F>* p) { zl::dbl_inplace(*p); }
// lane w: base_w = 2^(ZL_FB_BITS w) * G  (one-time, latency only)
template <class G>
__global__ void __launch_bounds__(64) k_fb_bases(XYZZ<typename G::F>* __restrict__ out) {
    using F = typename G::F;
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= ZL_FB_WINDOWS) return;
    XYZZ<F> p = XYZZ<F>::from_affine(Affine<F>{G::gen_x(), G::gen_y()});
    for (uint32_t k = 0; k < w * ZL_FB_BITS; k++) zl_dbl_ool(&p);
    out[w] = p;
}
// lane (w, d): d * base_w by double-and-add -> XYZZ (normalised afterwards by k_batch_affine)
template <class G>
__global__ void __launch_bounds__(64) k_fb_table(const XYZZ<typename G::F>* __restrict__ bases_w, XYZZ<typename G::F>* __restrict__ out) {
    using F = typename G::F;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)ZL_FB_WINDOWS << ZL_FB_BITS) return;
    const uint32_t w = t >> ZL_FB_BITS, d = t & ((1u << ZL_FB_BITS) - 1u);
    const XYZZ<F> base = bases_w[w];
    XYZZ<F> acc = XYZZ<F>::inf();
#pragma nounroll
    for (int i = ZL_FB_BITS - 1; i >= 0; i--) {
        zl_dbl_ool(&acc);
        if ((d >> i) & 1) zl_add_full_ool(&acc, &base);
    }
    out[t] = acc;
}
// out[i] = k[i] * G as XYZZ: one mixed addition per non-zero window digit
template <class G>
__global__ void __launch_bounds__(64) k_bases_generate_fb(const uint32_t* __restrict__ k, uint32_t n, const Affine<typename G::F>* __restrict__ table,
                                                           XYZZ<typename G::F>* __restrict__ out) {
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* sp = reinterpret_cast<const uint4*>(k + (size_t)i * 8);
    const uint4 lo = sp[0], hi = sp[1];
    const uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int w = 0; w < ZL_FB_WINDOWS; w++) {
        const uint32_t d = zl_get_bits(s, w * ZL_FB_BITS, ZL_FB_BITS);
        if (d == 0) continue;
        const Affine<F> P = table[((size_t)w << ZL_FB_BITS) + d];
        zl::add_mixed(acc, P.x, P.y, false);
    }
    out[i] = acc;
}
// batch_normalization_into_affine: lane t normalises elements t, t + lanes, t + 2 lanes, ... (coalesced) with ONE inversion: forward
// sweep stores the running product of the denominators, one Fermat inversion, backward sweep peels the inverses off.
// FORM 0: XYZZ (denominator zzz; x / zz, y / zzz), FORM 1: Jacobian (denominator z; x / z^2, y / z^3).  Infinity in -> infinity out.
template <class G, int FORM>
__global__ void __launch_bounds__(64) k_batch_affine(const void* __restrict__ in_, uint32_t n, uint32_t lanes, typename G::F* __restrict__ prefix,
                                                      Affine<typename G::F>* __restrict__ out) {
    using F = typename G::F;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lanes) return;
    const XYZZ<F>* in_x = reinterpret_cast<const XYZZ<F>*>(in_);
    const Jac<F>* in_j = reinterpret_cast<const Jac<F>*>(in_);
    F acc = F::one();
    uint32_t last = t;
    for (uint32_t i = t; i < n; i += lanes) {
        F d;
        if (FORM == 0) d = in_x[i].zzz; else d = in_j[i].z;
        prefix[i] = acc;
        if (!d.raw_zero()) acc = zl::mul(acc, d);
        last = i;
    }
    F u = zl::inv(acc);
    for (uint32_t i = last;; i -= lanes) {
        if (FORM == 0) {
            const XYZZ<F> p = in_x[i];
            if (p.is_inf()) {
                out[i] = Affine<F>::inf();
            } else {
                const F izzz = zl::mul(u, prefix[i]);
                u = zl::mul(u, p.zzz);
                const F tt = zl::mul(p.zz, izzz);  // 1 / z
                const F izz = zl::mul(tt, tt);     // 1 / zz
                out[i] = Affine<F>{zl::canon(zl::mul(p.x, izz)), zl::canon(zl::mul(p.y, izzz))};
            }
        } else {
            const Jac<F> p = in_j[i];
            if (p.is_inf()) {
                out[i] = Affine<F>::inf();
            } else {
                const F iz = zl::mul(u, prefix[i]);
                u = zl::mul(u, p.z);
                const F iz2 = zl::sqr(iz);
                out[i] = Affine<F>{zl::canon(zl::mul(p.x, iz2)), zl::canon(zl::mul(p.y, zl::mul(iz2, iz)))};
            }
        }
        if (i < lanes) break;
    }
}
// one level of the window table: out[i] = 2^c * in[i] in Jacobian coordinates (c doublings at 2M + 5S), normalised by k_batch_affine
template <class G>
__global__ void __launch_bounds__(64) k_bases_level_dbl(const Affine<typename G::F>* __restrict__ in, uint32_t n, int c, Jac<typename G::F>* __restrict__ out) {
    using F = typename G::F;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Jac<F> q = Jac<F>::from_affine(in[i]);
    for (int k = 0; k < c; k++) zl::jac_dbl_inplace(q);
    out[i] = q;
}
