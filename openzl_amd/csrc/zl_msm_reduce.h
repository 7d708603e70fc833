// zl_msm_reduce.h -- steps 4 and 5 of the MSM (zl_msm.hip): merge of the buckets cut by chunk boundaries, the sum of the scalar-1 bases and the
// bucket reduction sum_k k B_k (level-0 running sums + the channel tree).  Instantiated per group in zl_msm_tail.hip.
#pragma once
#include "zl_ctx.h"
#include <type_traits>
#include "zl_quad.h"
#include "zl_fq2pair.h"
#include "zl_msm_common.h"

// The merge / level-0 / tree kernels of an Fq2 group can compute in the inlining flavour of the field like the accumulation kernel
// (-DZL_HOT_TAILS): their out-of-line Fq2 product routines take 56 scalar arguments, 24 of which travel on the stack (236 - 1260 B of
// scratch per lane in round 2's G2 tails).
#ifdef ZL_HOT_TAILS
template <class F> using TailF = typename HotField<F>::type;
#else
template <class F> using TailF = F;
#endif
// one lane per bucket: empty -> infinity; cut into <= ZL_BIG_SPAN chunks -> fold partials; else defer to a block
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, (QUAD && sizeof(XYZZ<typename G::F>) <= 256) ? 3 : 1) k_msm_merge(const uint32_t* __restrict__ offsets, uint32_t NB, XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                   const XYZZ<typename G::F>* __restrict__ partials_, uint32_t* __restrict__ big_list,
                                                   uint32_t* __restrict__ big_count, uint32_t* __restrict__ giant_list, uint32_t* __restrict__ giant_count,
                                                   uint32_t ZL_CHUNK, uint32_t big_span) {
    ZL_SIDE_PRIO();
    using F = TailF<typename G::F>;
    XYZZ<F>* __restrict__ bucket_sums = reinterpret_cast<XYZZ<F>*>(bucket_sums_);
    const XYZZ<F>* __restrict__ partials = reinterpret_cast<const XYZZ<F>*>(partials_);
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = QUAD ? gt >> 2 : gt;  // QUAD: four lanes per bucket (zl_quad.h)
    const int sub = QUAD ? (int)(gt & 3u) : 0;
    if (b >= NB) return;
    const uint32_t s = offsets[b], e = offsets[b + 1];
    if (s == e) { if (sub == 0) bucket_sums[b] = XYZZ<F>::inf(); return; }
    const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
    if (t0 == t1) return;  // written directly by msm_accumulate
    if (t1 - t0 + 1 > ZL_GIANT_SPAN) { if (sub == 0) giant_list[atomicAdd(giant_count, 1u)] = b; return; }
    if (t1 - t0 + 1 > big_span) { if (sub == 0) big_list[atomicAdd(big_count, 1u)] = b; return; }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t t = t0; t <= t1; t++) {
        const XYZZ<F> p = partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)];
        if constexpr (QUAD) zl::add_full_quad(acc, p, sub);
        else zl::add_full(acc, p);
    }
    if (sub == 0) bucket_sums[b] = acc;
}
// The same merge with one lane per CHUNK BOUNDARY instead of one per bucket (round 6, launches that fill the machine): a bucket is cut where a chunk boundary
// p = t * ZL_CHUNK falls strictly inside it, and with tens of entries per bucket and 64-128 per chunk only every third to fourth bucket is -- in k_msm_merge
// three lanes of four leave at once and the additions of the rest run at a quarter of the wave (1.27 ms for 1.7 M additions at 2^24, c = 20: 1.3 additions
// per ns where the level-0 kernel sustains 5.5).  Here lane t looks up the bucket that holds entry p (the binary search the accumulation does per chunk) and
// folds its partials iff p is the FIRST boundary inside that bucket, so every lane that passes the two tests has exactly one bucket to fold.  Empty buckets
// are written by k_msm_fill_empty (memory-bound, no field arithmetic).  Same partials, same lists, same sums as k_msm_merge.
__device__ __forceinline__ uint32_t zl_bucket_of_entry(const uint32_t* __restrict__ offsets, uint32_t NB, uint32_t p) {
    uint32_t lo = 0, hi = NB + 1;  // first index with offsets[idx] > p, minus one: offsets[b] <= p < offsets[b + 1] (never an empty bucket)
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= p) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}
template <class G>
__global__ void __launch_bounds__(256) k_msm_fill_empty(const uint32_t* __restrict__ offsets, uint32_t NB, XYZZ<typename G::F>* __restrict__ bucket_sums) {
    ZL_SIDE_PRIO();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB || offsets[b] != offsets[b + 1]) return;
    bucket_sums[b] = XYZZ<typename G::F>::inf();
}
template <class G>
__global__ void __launch_bounds__(64, sizeof(XYZZ<typename G::F>) <= 256 ? 3 : 1) k_msm_merge_cuts(const uint32_t* __restrict__ offsets, uint32_t NB, XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                   const XYZZ<typename G::F>* __restrict__ partials_, uint32_t* __restrict__ big_list,
                                                   uint32_t* __restrict__ big_count, uint32_t* __restrict__ giant_list, uint32_t* __restrict__ giant_count,
                                                   uint32_t ZL_CHUNK, uint32_t big_span, uint32_t nchunks) {
    ZL_SIDE_PRIO();
    using F = TailF<typename G::F>;
    XYZZ<F>* __restrict__ bucket_sums = reinterpret_cast<XYZZ<F>*>(bucket_sums_);
    const XYZZ<F>* __restrict__ partials = reinterpret_cast<const XYZZ<F>*>(partials_);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x + 1;  // boundary between chunks t - 1 and t
    if (t >= nchunks) return;
    const uint64_t p64 = (uint64_t)t * ZL_CHUNK;
    if (p64 >= offsets[NB]) return;
    const uint32_t p = (uint32_t)p64;
    const uint32_t b = zl_bucket_of_entry(offsets, NB, p);
    const uint32_t s = offsets[b], e = offsets[b + 1];
    if (s == p || s / ZL_CHUNK != t - 1) return;  // not cut here, or an earlier boundary of the same bucket owns it
    const uint32_t t0 = t - 1, t1 = (e - 1) / ZL_CHUNK;
    if (t1 - t0 + 1 > ZL_GIANT_SPAN) { giant_list[atomicAdd(giant_count, 1u)] = b; return; }
    if (t1 - t0 + 1 > big_span) { big_list[atomicAdd(big_count, 1u)] = b; return; }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t tt = t0; tt <= t1; tt++) {
        const XYZZ<F> q = partials[(size_t)2 * tt + (s <= tt * ZL_CHUNK ? 0 : 1)];
        zl::add_full(acc, q);
    }
    bucket_sums[b] = acc;
}
// lanes of the block-tree kernels: 256 for G1, 128 for G2 (384-B points: a 256-lane block is capped at 256 VGPRs and spills)
template <class G>
struct TreeLanes { static constexpr int N = sizeof(XYZZ<typename G::F>) > 256 ? 128 : 256; };
template <class G>
__device__ __forceinline__ void zl_block_tree(XYZZ<typename G::F>* sh, XYZZ<typename G::F>& acc) {
    using F = typename G::F;
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = TreeLanes<G>::N / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            XYZZ<F> a = sh[threadIdx.x];
            const XYZZ<F> o = sh[threadIdx.x + off];
            zl::add_full(a, o);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    acc = sh[0];
    __syncthreads();
}
// one block per big bucket (ZL_BIG_SPAN < chunks <= ZL_GIANT_SPAN)
template <class G>
__global__ void __launch_bounds__(TreeLanes<G>::N) k_msm_merge_big(const uint32_t* __restrict__ offsets, XYZZ<typename G::F>* __restrict__ bucket_sums,
                                                        const XYZZ<typename G::F>* __restrict__ partials, const uint32_t* __restrict__ big_list,
                                                        const uint32_t* __restrict__ big_count, uint32_t ZL_CHUNK) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    for (uint32_t item = blockIdx.x; item < *big_count; item += gridDim.x) {
        const uint32_t b = big_list[item];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t t = t0 + threadIdx.x; t <= t1; t += blockDim.x) {
            const XYZZ<F> p = partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)];
            zl::add_full(acc, p);
        }
        zl_block_tree<G>(sh, acc);
        if (threadIdx.x == 0) bucket_sums[b] = acc;
    }
}
// giant buckets (> ZL_GIANT_SPAN chunks: many equal scalars), stage 1: block (item, part) tree-sums its share of the bucket's
// chunk partials -> giant_tmp[item * ZL_GIANT_PARTS + part]; stage 2: one lane per giant bucket folds the ZL_GIANT_PARTS sums
template <class G>
__global__ void __launch_bounds__(TreeLanes<G>::N) k_msm_merge_giant(const uint32_t* __restrict__ offsets, XYZZ<typename G::F>* __restrict__ giant_tmp,
                                                          const XYZZ<typename G::F>* __restrict__ partials, const uint32_t* __restrict__ giant_list,
                                                          const uint32_t* __restrict__ giant_count, uint32_t ZL_CHUNK) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    const uint32_t part = blockIdx.x % ZL_GIANT_PARTS;
    for (uint32_t item = blockIdx.x / ZL_GIANT_PARTS; item < *giant_count; item += gridDim.x / ZL_GIANT_PARTS) {
        const uint32_t b = giant_list[item];
        const uint32_t s = offsets[b], e = offsets[b + 1];
        const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
        const uint32_t per = (t1 - t0 + ZL_GIANT_PARTS) / ZL_GIANT_PARTS;  // ceil((t1 - t0 + 1) / parts)
        const uint32_t lo = t0 + part * per, hi = min(t1 + 1, lo + per);
        XYZZ<F> acc = XYZZ<F>::inf();
        for (uint32_t t = lo + threadIdx.x; t < hi; t += blockDim.x) {
            const XYZZ<F> p = partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)];
            zl::add_full(acc, p);
        }
        zl_block_tree<G>(sh, acc);
        if (threadIdx.x == 0) giant_tmp[(size_t)item * ZL_GIANT_PARTS + part] = acc;
    }
}
template <class G>
__global__ void __launch_bounds__(64) k_msm_merge_giant2(XYZZ<typename G::F>* __restrict__ bucket_sums, const XYZZ<typename G::F>* __restrict__ giant_tmp,
                                                          const uint32_t* __restrict__ giant_list, const uint32_t* __restrict__ giant_count) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= *giant_count) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t k = 0; k < ZL_GIANT_PARTS; k++) {
        const XYZZ<F> p = giant_tmp[(size_t)item * ZL_GIANT_PARTS + k];
        zl::add_full(acc, p);
    }
    bucket_sums[giant_list[item]] = acc;
}
// sum of the bases whose scalar is 1 (list built by the recoder): strided mixed adds per lane, block tree -> out[block]
#define ZL_ONES_BLOCKS 128

// Round 6: the kernel also FINISHES the sum (the last block to arrive folds the blocks' partial sums: no k_msm_window_sum launch behind it) and copies the job's two
// status words (entries accumulated, non-canonical-scalar flag) behind the result, so that one device-to-host copy fetches everything: two launches fewer on the
// latency-bound chain of a small MSM.  `done`: a per-job counter word, zero at launch.
template <class X>
__device__ __forceinline__ X zl_load_volatile(const X* p) {  // a value another workgroup of the same launch wrote (behind __threadfence + an atomic ticket)
    X r;
    const volatile uint32_t* s = reinterpret_cast<const volatile uint32_t*>(p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(X) / 4); i++) d[i] = s[i];
    return r;
}
template <class G>
__global__ void __launch_bounds__(TreeLanes<G>::N) k_msm_ones(const uint32_t* __restrict__ ones_list, const uint32_t* __restrict__ ones_count,
                                                   const Affine<typename G::F>* __restrict__ bases, XYZZ<typename G::F>* __restrict__ out,
                                                   const Affine<typename G::F>* __restrict__ phib, uint32_t n_real, XYZZ<typename G::F>* __restrict__ final_out,
                                                   uint32_t* __restrict__ done, const uint32_t* __restrict__ status_in) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    __shared__ uint32_t last_flag;
    const uint32_t cnt = *ones_count;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the status words travel with the result (final_out + 1: room for them in every result buffer)
        uint32_t* st = reinterpret_cast<uint32_t*>(final_out + 1);
        st[0] = status_in[0];
        st[1] = status_in[1];
    }
    if (cnt == 0) {  // the usual case for uniform scalars
        if (blockIdx.x == 0 && threadIdx.x == 0) *final_out = XYZZ<F>::inf();
        return;
    }
    const uint32_t nparts = min(gridDim.x, (cnt + blockDim.x - 1) / blockDim.x);  // blocks that have any index to add
    if (blockIdx.x >= nparts) return;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += nparts * blockDim.x) {
        const uint32_t idx = ones_list[j];
        const Affine<F> P = (G::GLV && idx >= n_real) ? phib[idx - n_real] : bases[idx];  // GLV: a half-scalar k2 = 1 names phi(P)
        if (!P.is_inf()) zl::add_mixed(acc, P.x, P.y, false);
    }
    if (cnt > 1) zl_block_tree<G>(sh, acc);  // (cnt == 1 -- a proof's constant ONE -- : lane 0 holds the point)
    if (nparts == 1) {
        if (threadIdx.x == 0) *final_out = acc;
        return;
    }
    if (threadIdx.x == 0) {
        out[blockIdx.x] = acc;
        __threadfence();
        last_flag = atomicAdd(done, 1u) == nparts - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last_flag) return;
    __threadfence();
    acc = XYZZ<F>::inf();
    for (uint32_t s = threadIdx.x; s < nparts; s += blockDim.x) {
        const XYZZ<F> p = zl_load_volatile(&out[s]);
        zl::add_full(acc, p);
    }
    zl_block_tree<G>(sh, acc);
    if (threadIdx.x == 0) *final_out = acc;
}

// ------------------------------------------------------------------------------------------------ bucket reduction
// sum_{k=1..H} k * B_k per bucket set without any scalar multiple and with a dependent chain of only ~log2(H) additions:
//   level 0   one lane per block of g0 consecutive buckets (bucket index i <-> weight i + 1): running sums in registers give
//             T = sum B_i and A = sum (i - i0 + 1) B_i   (2 additions per bucket, the classic trick inside the block)
//   tree      the remaining weight of block j is g0 * j.  sum_j j T_j = sum_b 2^b S_b with S_b = sum of the T_j whose index has bit b
//             set.  A binary tree over the block index carries, per node, the channels (T, A, S_0 .. S_(level-1)): combining children
//             (L, R) adds channel-wise, and the new top channel is S_(level-1) = T_R.  Every (node, channel) pair is ONE independent
//             addition (one lane), so a tree level is one launch of depth 1 and the whole reduction is log2(H / g0) dependent
//             additions -- no lane runs a double-and-add ladder for its offset and nothing is multiplied by a power of two on the
//             device.  Total work stays ~2 additions per bucket + ~3 per block.
//   host      the root's channels of every set, folded into the window Horner it runs anyway: position c w + log2 g0 + b receives
//             S_(w,b), position c w receives A_w (one extra addition per bit position, no extra doublings).
// flat_set: the spread top window (k_msm_recode_wide): its buckets are weighted by their low spread_t bits only; level 0 is weightless
// for it when spread_t == 0, and the host skips its S_b from bit spread_t on.
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, TreeLanes<G>::N == 128 ? 1 : 2) k_msm_reduce_level0(const XYZZ<typename G::F>* __restrict__ buckets_, uint32_t H, uint32_t group, uint32_t blocks_per_set,
                                                           uint32_t total_blocks, uint32_t flat_set, uint32_t flat_log,
                                                           XYZZ<typename G::F>* __restrict__ out_ /* [set][block][2]: T, A */) {
    ZL_SIDE_PRIO();
    using X = XYZZ<TailF<typename G::F>>;
    const X* __restrict__ buckets = reinterpret_cast<const X*>(buckets_);
    X* __restrict__ out = reinterpret_cast<X*>(out_);
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = QUAD ? gt >> 2 : gt;  // QUAD: four lanes per block of buckets (zl_quad.h)
    const int sub = QUAD ? (int)(gt & 3u) : 0;
    if (t >= total_blocks) return;
    const uint32_t set = t / blocks_per_set, blk = t % blocks_per_set;
    const uint32_t i0 = blk * group, i1 = min(H, i0 + group);
    const size_t base = (size_t)set * H;
    const bool flat = set == flat_set && flat_log == 0;
    X run = X::inf(), wsum = X::inf();
    for (uint32_t i = i1; i > i0; i--) {
        const X B = buckets[base + (i - 1)];
        if constexpr (QUAD) {
            zl::add_full_quad(run, B, sub);
            if (!flat) zl::add_full_quad(wsum, run, sub);
        } else {
            zl::add_full(run, B);
            if (!flat) zl::add_full(wsum, run);
        }
    }
    if (sub != 0) return;
    out[(size_t)2 * t] = run;
    // (two stores, not `flat ? run : wsum`: the conditional operator on the two structs becomes a select of their ADDRESSES, which pins both in scratch --
    // that was the whole of this kernel's 528 B of private memory in rounds 2-3)
    if (flat) out[(size_t)2 * t + 1] = run;
    else out[(size_t)2 * t + 1] = wsum;
}
// one tree level: nodes of `level` (1-based) from the nodes of level - 1.  Node layout: [set][node][channel], ch_in = level + 1 channels
// in (T, A, S_0 .. S_(level-2)), ch_out = level + 2 out.  One lane per (set, node, out channel).
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64) k_msm_reduce_tree(const XYZZ<typename G::F>* __restrict__ in_, XYZZ<typename G::F>* __restrict__ out_, uint32_t level,
                                                         uint32_t nodes_out_per_set, uint32_t total_lanes) {
    ZL_SIDE_PRIO();
    using X = XYZZ<TailF<typename G::F>>;
    const X* __restrict__ in = reinterpret_cast<const X*>(in_);
    X* __restrict__ out = reinterpret_cast<X*>(out_);
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = QUAD ? gt >> 2 : gt;  // QUAD: four lanes per (set, node, channel) (zl_quad.h)
    const int sub = QUAD ? (int)(gt & 3u) : 0;
    if (t >= total_lanes) return;
    const uint32_t ch_out = level + 2, ch_in = level + 1;
    const uint32_t ch = t % ch_out, node = (t / ch_out) % nodes_out_per_set, set = t / (ch_out * nodes_out_per_set);
    const size_t left = ((size_t)set * nodes_out_per_set * 2 + (size_t)2 * node) * ch_in, right = left + ch_in;
    if (ch == ch_out - 1) {  // the new top channel: blocks of the right child have this bit set
        if (sub == 0) out[t] = in[right];
        return;
    }
    X acc = in[left + ch];
    const X o = in[right + ch];
    if constexpr (QUAD) zl::add_full_quad(acc, o, sub);
    else zl::add_full(acc, o);
    if (sub == 0) out[t] = acc;
}
// ------------------------------------------------------------------------------------------------ Fq2 groups on lane pairs (zl_fq2pair.h)
// The same three kernels with TWO lanes per item (lane i of a row of 16: the c0 components, lane i ^ 8: the c1 components), for launches that fill the
// machine: the one-lane forms above hold three Fq2 points across out-of-line product calls -- 512 registers and 540-780 B of scratch per lane, one wave per
// SIMD -- where a half-point lane stays inside 256 registers with the product scans inlined.  Same buffers, same item numbering.
// QUAD: eight lanes per item (zl_fq2pair.h: four per half) for launches that do NOT fill the machine -- the chain of a small G2 MSM
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, 2) k_msm_merge_pair(const uint32_t* __restrict__ offsets, uint32_t NB, XYZZ<typename G::F>* __restrict__ bucket_sums,
                                                   const XYZZ<typename G::F>* __restrict__ partials, uint32_t* __restrict__ big_list,
                                                   uint32_t* __restrict__ big_count, uint32_t* __restrict__ giant_list, uint32_t* __restrict__ giant_count,
                                                   uint32_t ZL_CHUNK, uint32_t big_span) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        const int half = zl::pair_half(), sub = (int)(threadIdx.x & 3u);
        const uint32_t b = QUAD ? ZL_OCTET_ITEM() : ZL_PAIR_ITEM();
        if (b >= NB) return;
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (s == e) { pair_store(&bucket_sums[b], half, X::inf()); return; }
        const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
        if (t0 == t1) return;
        const bool first = half == 0 && (!QUAD || sub == 0);
        if (t1 - t0 + 1 > ZL_GIANT_SPAN) { if (first) giant_list[atomicAdd(giant_count, 1u)] = b; return; }
        if (t1 - t0 + 1 > big_span) { if (first) big_list[atomicAdd(big_count, 1u)] = b; return; }
        X acc = X::inf();
        for (uint32_t t = t0; t <= t1; t++) {
            const X p = pair_load(&partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)], half);
            if constexpr (QUAD) zl::add_full_quad(acc, p, sub);
            else zl::add_full(acc, p);
        }
        pair_store(&bucket_sums[b], half, acc);
    }
}
template <class G>
__global__ void __launch_bounds__(64, 2) k_msm_merge_cuts_pair(const uint32_t* __restrict__ offsets, uint32_t NB, XYZZ<typename G::F>* __restrict__ bucket_sums,
                                                   const XYZZ<typename G::F>* __restrict__ partials, uint32_t* __restrict__ big_list,
                                                   uint32_t* __restrict__ big_count, uint32_t* __restrict__ giant_list, uint32_t* __restrict__ giant_count,
                                                   uint32_t ZL_CHUNK, uint32_t big_span, uint32_t nchunks) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        const int half = zl::pair_half();
        const uint32_t t = ZL_PAIR_ITEM() + 1;  // (k_msm_merge_cuts: one lane PAIR per chunk boundary)
        if (t >= nchunks) return;
        const uint64_t p64 = (uint64_t)t * ZL_CHUNK;
        if (p64 >= offsets[NB]) return;
        const uint32_t p = (uint32_t)p64;
        const uint32_t b = zl_bucket_of_entry(offsets, NB, p);
        const uint32_t s = offsets[b], e = offsets[b + 1];
        if (s == p || s / ZL_CHUNK != t - 1) return;
        const uint32_t t0 = t - 1, t1 = (e - 1) / ZL_CHUNK;
        if (t1 - t0 + 1 > ZL_GIANT_SPAN) { if (half == 0) giant_list[atomicAdd(giant_count, 1u)] = b; return; }
        if (t1 - t0 + 1 > big_span) { if (half == 0) big_list[atomicAdd(big_count, 1u)] = b; return; }
        X acc = X::inf();
        for (uint32_t tt = t0; tt <= t1; tt++) {
            const X q = pair_load(&partials[(size_t)2 * tt + (s <= tt * ZL_CHUNK ? 0 : 1)], half);
            zl::add_full(acc, q);
        }
        pair_store(&bucket_sums[b], half, acc);
    }
}
#ifndef ZL_L0_PAIR_WAVES
#define ZL_L0_PAIR_WAVES 2  // (1 = no scratch at one wave per SIMD: the A/B of profiles/r06_l0_pair_waves_ab.log)
#endif
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, ZL_L0_PAIR_WAVES) k_msm_reduce_level0_pair(const XYZZ<typename G::F>* __restrict__ buckets, uint32_t H, uint32_t group, uint32_t blocks_per_set,
                                                           uint32_t total_blocks, uint32_t flat_set, uint32_t flat_log, XYZZ<typename G::F>* __restrict__ out) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        const int half = zl::pair_half(), sub = (int)(threadIdx.x & 3u);
        const uint32_t t = QUAD ? ZL_OCTET_ITEM() : ZL_PAIR_ITEM();
        if (t >= total_blocks) return;
        const uint32_t set = t / blocks_per_set, blk = t % blocks_per_set;
        const uint32_t i0 = blk * group, i1 = min(H, i0 + group);
        const size_t base = (size_t)set * H;
        const bool flat = set == flat_set && flat_log == 0;
        X run = X::inf(), wsum = X::inf();
        for (uint32_t i = i1; i > i0; i--) {
            const X Bk = pair_load(&buckets[base + (i - 1)], half);
            if constexpr (QUAD) {
                zl::add_full_quad(run, Bk, sub);
                if (!flat) zl::add_full_quad(wsum, run, sub);
            } else {
                zl::add_full(run, Bk);
                if (!flat) zl::add_full(wsum, run);
            }
        }
        pair_store(&out[(size_t)2 * t], half, run);
        if (flat) pair_store(&out[(size_t)2 * t + 1], half, run);
        else pair_store(&out[(size_t)2 * t + 1], half, wsum);
    }
}
template <class G, bool QUAD = false>
__global__ void __launch_bounds__(64, 2) k_msm_reduce_tree_pair(const XYZZ<typename G::F>* __restrict__ in, XYZZ<typename G::F>* __restrict__ out, uint32_t level,
                                                         uint32_t nodes_out_per_set, uint32_t total_lanes) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        const int half = zl::pair_half(), sub = (int)(threadIdx.x & 3u);
        const uint32_t t = QUAD ? ZL_OCTET_ITEM() : ZL_PAIR_ITEM();
        if (t >= total_lanes) return;
        const uint32_t ch_out = level + 2, ch_in = level + 1;
        const uint32_t ch = t % ch_out, node = (t / ch_out) % nodes_out_per_set, set = t / (ch_out * nodes_out_per_set);
        const size_t left = ((size_t)set * nodes_out_per_set * 2 + (size_t)2 * node) * ch_in, right = left + ch_in;
        if (ch == ch_out - 1) {
            pair_store(&out[t], half, pair_load(&in[right], half));
            return;
        }
        X acc = pair_load(&in[left + ch], half);
        const X o = pair_load(&in[right + ch], half);
        if constexpr (QUAD) zl::add_full_quad(acc, o, sub);
        else zl::add_full(acc, o);
        pair_store(&out[t], half, acc);
    }
}
// The block-tree kernels on lane pairs (round 6): the heavy buckets of a skewed input -- and a Groth16 witness IS skewed: the table-mode G2 MSM of the 958 465-constraint
// proof spent 0.56 ms in k_msm_merge_big<BlsG2> and 1.3 ms in k_msm_merge_giant<BlsG2> per proof, one-lane kernels at 512 registers + 716 B of scratch
// (profiles/r06_kernel_stats_groth16.txt).  A block of 2 N lanes holds N items (ZL_PAIR_ITEM_IN_BLOCK); the LDS tree stores each item as the XYZZ<Fq2> it is in
// global memory, every lane moving the component of its half.
#define ZL_PAIR_ITEM_IN_BLOCK() (((threadIdx.x >> 4) << 3) + (threadIdx.x & 7u))
template <class G, int N>
__device__ __forceinline__ void zl_block_tree_pair(XYZZ<typename G::F>* sh, XYZZ<Fp2H<typename PairBase<typename G::F>::type>>& acc, uint32_t item, int half) {
    using X = XYZZ<Fp2H<typename PairBase<typename G::F>::type>>;
    pair_store(&sh[item], half, acc);
    __syncthreads();
    for (uint32_t off = N / 2; off > 0; off >>= 1) {
        if (item < off) {
            X a = pair_load(&sh[item], half);
            const X o = pair_load(&sh[item + off], half);
            zl::add_full(a, o);
            pair_store(&sh[item], half, a);
        }
        __syncthreads();
    }
    acc = pair_load(&sh[0], half);
    __syncthreads();
}
template <class G>
__global__ void __launch_bounds__(2 * TreeLanes<G>::N) k_msm_merge_big_pair(const uint32_t* __restrict__ offsets, XYZZ<typename G::F>* __restrict__ bucket_sums,
                                                        const XYZZ<typename G::F>* __restrict__ partials, const uint32_t* __restrict__ big_list,
                                                        const uint32_t* __restrict__ big_count, uint32_t ZL_CHUNK) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        constexpr uint32_t N = TreeLanes<G>::N;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        XYZZ<typename G::F>* sh = reinterpret_cast<XYZZ<typename G::F>*>(smem);
        const int half = zl::pair_half();
        const uint32_t item = ZL_PAIR_ITEM_IN_BLOCK();
        for (uint32_t it = blockIdx.x; it < *big_count; it += gridDim.x) {
            const uint32_t b = big_list[it];
            const uint32_t s = offsets[b], e = offsets[b + 1];
            const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
            X acc = X::inf();
            for (uint32_t t = t0 + item; t <= t1; t += N) {
                const X p = pair_load(&partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)], half);
                zl::add_full(acc, p);
            }
            zl_block_tree_pair<G, N>(sh, acc, item, half);
            if (item == 0) pair_store(&bucket_sums[b], half, acc);
        }
    }
}
template <class G>
__global__ void __launch_bounds__(2 * TreeLanes<G>::N) k_msm_merge_giant_pair(const uint32_t* __restrict__ offsets, XYZZ<typename G::F>* __restrict__ giant_tmp,
                                                          const XYZZ<typename G::F>* __restrict__ partials, const uint32_t* __restrict__ giant_list,
                                                          const uint32_t* __restrict__ giant_count, uint32_t ZL_CHUNK) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        constexpr uint32_t N = TreeLanes<G>::N;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        XYZZ<typename G::F>* sh = reinterpret_cast<XYZZ<typename G::F>*>(smem);
        const int half = zl::pair_half();
        const uint32_t item = ZL_PAIR_ITEM_IN_BLOCK();
        const uint32_t part = blockIdx.x % ZL_GIANT_PARTS;
        for (uint32_t it = blockIdx.x / ZL_GIANT_PARTS; it < *giant_count; it += gridDim.x / ZL_GIANT_PARTS) {
            const uint32_t b = giant_list[it];
            const uint32_t s = offsets[b], e = offsets[b + 1];
            const uint32_t t0 = s / ZL_CHUNK, t1 = (e - 1) / ZL_CHUNK;
            const uint32_t per = (t1 - t0 + ZL_GIANT_PARTS) / ZL_GIANT_PARTS;
            const uint32_t lo = t0 + part * per, hi = min(t1 + 1, lo + per);
            X acc = X::inf();
            for (uint32_t t = lo + item; t < hi; t += N) {
                const X p = pair_load(&partials[(size_t)2 * t + (s <= t * ZL_CHUNK ? 0 : 1)], half);
                zl::add_full(acc, p);
            }
            zl_block_tree_pair<G, N>(sh, acc, item, half);
            if (item == 0) pair_store(&giant_tmp[(size_t)it * ZL_GIANT_PARTS + part], half, acc);
        }
    }
}
// stage 2: one block of ZL_GIANT_PARTS pairs per giant bucket (the one-lane kernel folds the 32 sums serially: 32 Fq2 additions of latency)
template <class G>
__global__ void __launch_bounds__(2 * ZL_GIANT_PARTS) k_msm_merge_giant2_pair(XYZZ<typename G::F>* __restrict__ bucket_sums, XYZZ<typename G::F>* __restrict__ giant_tmp,
                                                          const uint32_t* __restrict__ giant_list, const uint32_t* __restrict__ giant_count) {
    using B = typename PairBase<typename G::F>::type;
    if constexpr (!std::is_void<B>::value) {
        ZL_SIDE_PRIO();
        using X = XYZZ<Fp2H<B>>;
        const int half = zl::pair_half();
        const uint32_t item = ZL_PAIR_ITEM_IN_BLOCK();
        for (uint32_t it = blockIdx.x; it < *giant_count; it += gridDim.x) {
            XYZZ<typename G::F>* sh = giant_tmp + (size_t)it * ZL_GIANT_PARTS;  // the tree runs in place over the bucket's 32 part sums (global memory: 16 KB, once per giant bucket)
            X acc = pair_load(&sh[item], half);
            __syncthreads();
            zl_block_tree_pair<G, ZL_GIANT_PARTS>(sh, acc, item, half);
            if (item == 0) {
                pair_store(&bucket_sums[giant_list[it]], half, acc);
            }
        }
    }
}
// tree-sum of segment results.  Block b belongs to set (b / parts) and sums `count` consecutive elements starting at
// set * set_stride + (b % parts) * count (clipped to the set): parts = 1 -> one block per set; parts > 1 -> stage 1 of a
// two-stage sum for sets with many segments.
template <class G>
__global__ void __launch_bounds__(TreeLanes<G>::N) k_msm_window_sum(const XYZZ<typename G::F>* __restrict__ seg_out, uint32_t count, uint32_t set_stride,
                                                         uint32_t parts, XYZZ<typename G::F>* __restrict__ out, const uint32_t* __restrict__ zero_if_zero) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    if (zero_if_zero && *zero_if_zero == 0) {  // nothing was listed: every part is the point at infinity
        if (threadIdx.x == 0) out[blockIdx.x] = XYZZ<F>::inf();
        return;
    }
    const uint32_t set = blockIdx.x / parts, part = blockIdx.x % parts;
    const uint32_t lo = part * count, hi = min(set_stride, lo + count);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t s = lo + threadIdx.x; s < hi; s += blockDim.x) {
        const XYZZ<F> p = seg_out[(size_t)set * set_stride + s];
        zl::add_full(acc, p);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int off = TreeLanes<G>::N / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            XYZZ<F> a = sh[threadIdx.x];
            const XYZZ<F> o = sh[threadIdx.x + off];
            zl::add_full(a, o);
            sh[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// every instantiation MsmJob<G>::tail launches (X as in zl_msm_accumulate.h)
#define ZL_MSM_TAIL_KERNELS(X, G) \
    X template __global__ void k_msm_merge<G, false>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t); \
    X template __global__ void k_msm_merge<G, true>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t); \
    X template __global__ void k_msm_merge_big<G>(const uint32_t*, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, const uint32_t*, const uint32_t*, uint32_t); \
    X template __global__ void k_msm_merge_giant<G>(const uint32_t*, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, const uint32_t*, const uint32_t*, uint32_t); \
    X template __global__ void k_msm_merge_giant2<G>(XYZZ<typename G::F>*, const XYZZ<typename G::F>*, const uint32_t*, const uint32_t*); \
    X template __global__ void k_msm_ones<G>(const uint32_t*, const uint32_t*, const Affine<typename G::F>*, XYZZ<typename G::F>*, const Affine<typename G::F>*, uint32_t, XYZZ<typename G::F>*, uint32_t*, const uint32_t*); \
    X template __global__ void k_msm_reduce_level0<G, false>(const XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, XYZZ<typename G::F>*); \
    X template __global__ void k_msm_reduce_level0<G, true>(const XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, XYZZ<typename G::F>*); \
    X template __global__ void k_msm_reduce_tree<G, false>(const XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_reduce_tree<G, true>(const XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_fill_empty<G>(const uint32_t*, uint32_t, XYZZ<typename G::F>*); \
    X template __global__ void k_msm_merge_cuts<G>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_merge_cuts_pair<G>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_merge_big_pair<G>(const uint32_t*, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, const uint32_t*, const uint32_t*, uint32_t); \
    X template __global__ void k_msm_merge_giant_pair<G>(const uint32_t*, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, const uint32_t*, const uint32_t*, uint32_t); \
    X template __global__ void k_msm_merge_giant2_pair<G>(XYZZ<typename G::F>*, XYZZ<typename G::F>*, const uint32_t*, const uint32_t*); \
    X template __global__ void k_msm_merge_pair<G, false>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t); \
    X template __global__ void k_msm_merge_pair<G, true>(const uint32_t*, uint32_t, XYZZ<typename G::F>*, const XYZZ<typename G::F>*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t); \
    X template __global__ void k_msm_reduce_level0_pair<G, false>(const XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, XYZZ<typename G::F>*); \
    X template __global__ void k_msm_reduce_level0_pair<G, true>(const XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, XYZZ<typename G::F>*); \
    X template __global__ void k_msm_reduce_tree_pair<G, false>(const XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_reduce_tree_pair<G, true>(const XYZZ<typename G::F>*, XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t); \
    X template __global__ void k_msm_window_sum<G>(const XYZZ<typename G::F>*, uint32_t, uint32_t, uint32_t, XYZZ<typename G::F>*, const uint32_t*);
