// zl_msm_accumulate.h -- step 3 of the MSM (zl_msm.hip): the bucket accumulation kernels, one lane (or one DPP quad) per chunk of the
// bucket-sorted entry list.  Instantiated per group in zl_msm_acc.hip; zl_msm.hip only launches them (ZL_MSM_ACCUMULATE_KERNELS(extern, G)).
#pragma once
#include <type_traits>
#include "zl_ctx.h"
#include "zl_quad.h"
#include "zl_fq2pair.h"
#include "zl_msm_common.h"

// ------------------------------------------------------------------------------------------------ accumulate
__device__ __forceinline__ uint32_t zl_upper_bound(const uint32_t* __restrict__ a, uint32_t n, uint32_t key) {
    // first index with a[idx] > key
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (a[mid] <= key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#ifndef ZL_ACC_WAVES
#define ZL_ACC_WAVES 2  // waves per SIMD the accumulate kernel is register-budgeted for
#endif
#ifndef ZL_ACC_BLOCK
#define ZL_ACC_BLOCK 64
#endif
// one lane, one chunk: entries [t * ZL_CHUNK, (t + 1) * ZL_CHUNK) of the bucket-sorted list
template <class G, bool QUAD = false>
__device__ __forceinline__ void zl_accumulate_chunk(uint32_t t, int sub, uint32_t E, const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                    const Affine<typename HotField<typename G::F>::type>* __restrict__ bases,
                                                    XYZZ<typename HotField<typename G::F>::type>* __restrict__ bucket_sums,
                                                    XYZZ<typename HotField<typename G::F>::type>* __restrict__ partials, uint32_t ZL_CHUNK,
                                                    const Affine<typename HotField<typename G::F>::type>* __restrict__ phib, uint32_t n_real, uint32_t idx_mask = 0x7fffffffu) {
    using F = typename HotField<typename G::F>::type;  // same layout as G::F; Fq2 on 28-bit limbs: the inlining flavour (zl_curve.h)
    const uint64_t start64 = (uint64_t)t * ZL_CHUNK;
    if (start64 >= E) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (uint32_t)min((uint64_t)E, start64 + ZL_CHUNK);
    uint32_t b = zl_upper_bound(offsets, NB + 1, start) - 1;  // bucket holding entry `start`
    uint32_t b_start = offsets[b], b_end = offsets[b + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    // ONE flat loop of exactly (end - start) mixed additions per lane: a per-segment inner loop would make the
    // wave run max-over-lanes iterations per segment (measured 2.4x slower).  Bucket boundaries only flush.
    for (uint32_t e = start; e < end; e++) {
        while (e == b_end) {  // lane crosses into the next bucket (empty buckets: zero-length, skipped here)
            if (b_end > b_start) {
                if (!QUAD || sub == 0) {
                    if (b_start >= start) bucket_sums[b] = acc;  // bucket lies inside this chunk (b_end <= e < end)
                    else partials[(size_t)2 * t] = acc;          // head bucket started in an earlier chunk
                }
                acc = XYZZ<F>::inf();
            }
            b++;
            b_start = b_end;
            b_end = offsets[b + 1];
        }
        const uint32_t ent = entries[e];
        const uint32_t idx = ent & idx_mask;  // (the product kernels pass the literal: the sign bit off; the measurement kernel may fold the gather into a cache-resident prefix)
        const Affine<F> P = (G::GLV && idx >= n_real ? phib : bases)[idx];
        if (!P.is_inf()) {
            if constexpr (QUAD) zl::add_mixed_quad(acc, P.x, P.y, (ent >> 31) != 0, sub);
            else zl::add_mixed(acc, P.x, P.y, (ent >> 31) != 0);
        }
    }
    // last segment [max(b_start,start), end) of bucket b
    const bool complete = (b_start >= start) && (b_end <= end);
    if (QUAD && sub != 0) return;
    if (complete) bucket_sums[b] = acc;
    else partials[(size_t)2 * t + (b_start <= start ? 0 : 1)] = acc;
}
template <class G>
__global__ void __launch_bounds__(ZL_ACC_BLOCK, ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK > 0 ? ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK : 1) k_msm_accumulate(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                        const Affine<typename G::F>* __restrict__ bases_,
                                                        XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                        XYZZ<typename G::F>* __restrict__ partials_, uint32_t ZL_CHUNK,
                                                        const Affine<typename G::F>* __restrict__ phib_, uint32_t n_real) {
    using F = typename HotField<typename G::F>::type;
    static_assert(sizeof(F) == sizeof(typename G::F), "hot flavour must share the layout");
    // GLV: virtual point n_real + i = phi(P_i); else n_real = 2^32 - 1 (never selected)
    zl_accumulate_chunk<G>(blockIdx.x * blockDim.x + threadIdx.x, 0, offsets[NB], entries, offsets, NB, reinterpret_cast<const Affine<F>*>(bases_),
                           reinterpret_cast<XYZZ<F>*>(bucket_sums_), reinterpret_cast<XYZZ<F>*>(partials_), ZL_CHUNK,
                           reinterpret_cast<const Affine<F>*>(phib_) - n_real, n_real);
}
#ifdef ZL_MEASURE
// MEASUREMENT BUILDS ONLY (-DZL_MEASURE: ZL_EXTRA_FLAGS=-DZL_MEASURE ZL_BUILD_TAG=measure python -m openzl_amd.build; zl_test_acc_clock, include/zl_backend_test.h): the same kernel with four scalar clock reads per WAVE -- s_memtime (shader cycles) and
// s_memrealtime (the constant 100 MHz counter) at its start and at its end -- so that the effective shader clock of the accumulation (the chip clocks
// dense VALU bodies to its power budget, MI355X_MICROARCH.md "DVFS give-back") is read from the kernel itself: sum of cycle deltas / sum of tick deltas.
// One record of four u64 per workgroup (= wave).  G1 groups only; never launched by the product path unless the hook armed ctx->acc_clk.
// idx_mask: 0x7fffffff = the product's gather; ZL_TUNE_ACC_CLK_IDX_BITS=b folds every base index into the first 2^b points (WRONG sums, measurement only): the
// same arithmetic with the 128-B gathers served from L2 / MALL instead of HBM -- what the gathers cost in time and in clock (profiles/r05_acc_gather_ab.log).
template <class G>
__global__ void __launch_bounds__(ZL_ACC_BLOCK, ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK > 0 ? ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK : 1) k_msm_accumulate_clk(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                        const Affine<typename G::F>* __restrict__ bases_,
                                                        XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                        XYZZ<typename G::F>* __restrict__ partials_, uint32_t ZL_CHUNK,
                                                        const Affine<typename G::F>* __restrict__ phib_, uint32_t n_real, unsigned long long* __restrict__ clk, uint32_t idx_mask) {
    if constexpr (G::COORDS == 1) {
        using F = typename HotField<typename G::F>::type;
        const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
        zl_accumulate_chunk<G>(blockIdx.x * blockDim.x + threadIdx.x, 0, offsets[NB], entries, offsets, NB, reinterpret_cast<const Affine<F>*>(bases_),
                               reinterpret_cast<XYZZ<F>*>(bucket_sums_), reinterpret_cast<XYZZ<F>*>(partials_), ZL_CHUNK,
                               reinterpret_cast<const Affine<F>*>(phib_) - n_real, n_real, idx_mask);
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            unsigned long long* o = clk + (size_t)4 * blockIdx.x;
            o[0] = c0; o[1] = c1; o[2] = w0; o[3] = w1;
        }
    }
}
#endif
// The same chunks with FOUR lanes per chunk (zl_quad.h): for lists that do not fill the machine (small MSMs), where the time of the launch is
// the latency of one lane's chain of mixed additions -- 4 product slots per addition instead of 10.5.
template <class G>
__global__ void __launch_bounds__(ZL_ACC_BLOCK, ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK > 0 ? ZL_ACC_WAVES * 64 / ZL_ACC_BLOCK : 1) k_msm_accumulate_quad(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                        const Affine<typename G::F>* __restrict__ bases_,
                                                        XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                        XYZZ<typename G::F>* __restrict__ partials_, uint32_t ZL_CHUNK,
                                                        const Affine<typename G::F>* __restrict__ phib_, uint32_t n_real) {
    using F = typename HotField<typename G::F>::type;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    zl_accumulate_chunk<G, true>(gt >> 2, (int)(gt & 3u), offsets[NB], entries, offsets, NB, reinterpret_cast<const Affine<F>*>(bases_),
                                 reinterpret_cast<XYZZ<F>*>(bucket_sums_), reinterpret_cast<XYZZ<F>*>(partials_), ZL_CHUNK,
                                 reinterpret_cast<const Affine<F>*>(phib_) - n_real, n_real);
}
// Fq2 groups, TWO lanes per chunk (zl_fq2pair.h): lane i of a row of 16 holds the c0 components of the chunk's running sum and of the base it adds,
// lane i ^ 8 the c1 components; a wave walks 32 chunks.  Registers per lane halve (416 -> two waves per SIMD), the instruction stream of one mixed
// addition halves (5.3 k mads: it fits the instruction cache), the mads per addition stay the same.  Same chunks, same buckets, same partials as
// zl_accumulate_chunk: the merge and reduction kernels do not know which of the two wrote them.
#ifndef ZL_ACC_PAIR_WAVES
#define ZL_ACC_PAIR_WAVES 2
#endif
// QUAD: eight lanes per chunk (four per half) for lists that do not fill the machine
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, ZL_ACC_PAIR_WAVES) k_msm_accumulate_pair(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                        const Affine<typename G::F>* __restrict__ bases_,
                                                        XYZZ<typename G::F>* __restrict__ bucket_sums,
                                                        XYZZ<typename G::F>* __restrict__ partials, uint32_t ZL_CHUNK,
                                                        const Affine<typename G::F>* __restrict__ phib_, uint32_t n_real) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        using H = Fp2H<B>;
        const int half = zl::pair_half(), sub = (int)(threadIdx.x & 3u);
        const uint32_t t = QUAD ? ZL_OCTET_ITEM() : ZL_PAIR_ITEM();
        const uint32_t E = offsets[NB];
        const Affine<typename G::F>* __restrict__ phib = phib_ - n_real;  // GLS: virtual point n_real + i = psi^j(P_i); else n_real = 2^32 - 1 (never selected)
        const uint64_t start64 = (uint64_t)t * ZL_CHUNK;
        if (start64 >= E) return;
        const uint32_t start = (uint32_t)start64;
        const uint32_t end = (uint32_t)min((uint64_t)E, start64 + ZL_CHUNK);
        uint32_t b = zl_upper_bound(offsets, NB + 1, start) - 1;
        uint32_t b_start = offsets[b], b_end = offsets[b + 1];
        XYZZ<H> acc = XYZZ<H>::inf();
        for (uint32_t e = start; e < end; e++) {  // (the flat loop of zl_accumulate_chunk)
            while (e == b_end) {
                if (b_end > b_start) {
                    if (b_start >= start) pair_store(&bucket_sums[b], half, acc);
                    else pair_store(&partials[(size_t)2 * t], half, acc);
                    acc = XYZZ<H>::inf();
                }
                b++;
                b_start = b_end;
                b_end = offsets[b + 1];
            }
            const uint32_t ent = entries[e];
            const uint32_t idx = ent & 0x7fffffffu;
            const Affine<H> P = pair_load(&(G::GLV && idx >= n_real ? phib : bases_)[idx], half);
            if (!P.is_inf()) {
                if constexpr (QUAD) zl::add_mixed_quad(acc, P.x, P.y, (ent >> 31) != 0, sub);
                else zl::add_mixed(acc, P.x, P.y, (ent >> 31) != 0);
            }
        }
        const bool complete = (b_start >= start) && (b_end <= end);
        if (complete) pair_store(&bucket_sums[b], half, acc);
        else pair_store(&partials[(size_t)2 * t + (b_start <= start ? 0 : 1)], half, acc);
    }
}
// every instantiation MsmJob<G>::accumulate launches: X = empty defines them (zl_msm_acc.hip), X = extern only declares them
#ifdef ZL_MEASURE
#define ZL_MSM_ACCUMULATE_CLK_KERNEL(X, G) X template __global__ void k_msm_accumulate_clk<G>(const uint32_t*, const uint32_t*, uint32_t, const Affine<typename G::F>*, XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, const Affine<typename G::F>*, uint32_t, unsigned long long*, uint32_t);
#else
#define ZL_MSM_ACCUMULATE_CLK_KERNEL(X, G)
#endif
#define ZL_MSM_ACCUMULATE_KERNELS(X, G) \
    ZL_MSM_ACCUMULATE_CLK_KERNEL(X, G) \
    X template __global__ void k_msm_accumulate<G>(const uint32_t*, const uint32_t*, uint32_t, const Affine<typename G::F>*, XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, const Affine<typename G::F>*, uint32_t); \
    X template __global__ void k_msm_accumulate_quad<G>(const uint32_t*, const uint32_t*, uint32_t, const Affine<typename G::F>*, XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, const Affine<typename G::F>*, uint32_t); \
    X template __global__ void k_msm_accumulate_pair<G, true>(const uint32_t*, const uint32_t*, uint32_t, const Affine<typename G::F>*, XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, const Affine<typename G::F>*, uint32_t); \
    X template __global__ void k_msm_accumulate_pair<G, false>(const uint32_t*, const uint32_t*, uint32_t, const Affine<typename G::F>*, XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, const Affine<typename G::F>*, uint32_t); \

