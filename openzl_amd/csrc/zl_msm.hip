// zl_msm.hip -- variable-base multi-scalar multiplication on gfx950 (Pippenger, bucket method).
//
// Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0; reached from
// /root/reference/plugins/arkworks/src/groth16.rs:454 through ark-groth16's create_proof).
// arkworks walks the windows serially with unsigned c-bit digits and 2^c - 1 Jacobian buckets per window.  This
// backend is organised for a 256-CU / wave64 machine instead:
//   1. recode         signed-digit recoding of every scalar, once (2^(c-1) buckets per window); scalars equal to 1 go to a compact list,
//                     scalars of bases at infinity are dropped
//   2. sort           (window, bucket) counting sort of (point index, sign) entries without global atomics on the streaming paths:
//                     c <= 16: per-(slice, window) LDS histograms + range-owned scatter; c = 17..20 and the table mode: three levels --
//                     64..208 groups -> 128 sub-groups of 256 buckets -> one LDS-staged sort per sub-group (oversized sub-groups in tiles)
//   3. msm_accumulate the hot kernel: the sorted entry array is cut into fixed chunks of <= 128 entries, one lane per chunk, XYZZ mixed
//                     additions in ONE flat loop; a bucket inside one chunk is written directly, buckets cut by chunk boundaries leave
//                     <= 2 partial sums per lane -> perfect load balance whatever the scalar distribution (Groth16 witnesses are full
//                     of 0/1)
//   4. msm_merge      one lane per bucket folds the partials of cut buckets (block-wide tree for big / giant buckets)
//   5. reduce         sum_k k*B_k per bucket set: running sums over blocks of 8 buckets, then a binary tree whose nodes carry per-bit channel
//                     sums (one independent addition per lane and level, no scalar multiples)
//   6. host           ONE Horner over the bit positions of the scalar absorbs the channel sums and the window weights (~500 group
//                     operations), returned as an XYZZ partial
// Plain calls pick c from n (19-20 bits at 2^24: 13-14 additions per point).  With zl_bases_precompute (table of 2^(c w) P_i, W x the
// memory) all windows share ONE bucket set and c grows to 22 (12 additions per point).
// The result does not depend on c, on the digit signs or on the order entries land in a bucket (group law).
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <vector>
#include "zl_ctx.h"
#include "zl_pool.h"
#include "zl_quad.h"

// This file is compiled once per group: -DZL_G=BlsG1|BnG1|BlsG2|BnG2 (see openzl_amd/build.py)
#ifndef ZL_G
#error "compile with -DZL_G=<group config>"
#endif
#define ZL_GCAT_(a, b) a##_##b
#define ZL_GCAT(a, b) ZL_GCAT_(a, b)
#define ZL_GNAME(f) ZL_GCAT(f, ZL_G)

// Wave issue priority of the sort / tail kernels: in a pipeline they share every SIMD with two waves of the accumulation kernel, which would
// otherwise win most issue slots (a 1-ms sort kernel then takes 5-9 ms); their own VALU demand is tiny.
#ifdef ZL_NO_SIDE_PRIO
#define ZL_SIDE_PRIO() ((void)0)
#else
#define ZL_SIDE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
#define ZL_CHUNK_MAX 64    // entries per lane in msm_accumulate (smaller for small inputs: more lanes, shorter chains)
#define ZL_BIG_SPAN 64     // buckets cut into more chunks than this are merged by a whole block ...
#define ZL_BIG_SPAN_SMALL 8  // ... 8 for small inputs: a lane folds its partials serially, and 64 dependent additions (1.2 ms for G1, 3 ms
                             // for G2) were the whole tail of a small Groth16 proof; for large inputs the serial fold is the cheaper one
#define ZL_GIANT_SPAN 4096 // ... and into more than this by ZL_GIANT_PARTS blocks (two stages)
#define ZL_GIANT_PARTS 32


// ------------------------------------------------------------------------------------------------ digits
__device__ __forceinline__ uint32_t zl_get_bits(const uint32_t* __restrict__ s, int pos, int c) {
    // bits [pos, pos+c) of a 256-bit little-endian integer (c <= 24); bits above 255 read as 0
    int word = pos >> 5, sh = pos & 31;
    if (word >= 8) return 0;
    uint64_t v = s[word];
    if (word + 1 < 8) v |= (uint64_t)s[word + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1);
}


// Scalars equal to 1 (boolean witnesses: a large share of a Groth16 assignment; arkworks' MSM special-cases them too) bypass
// the sort: the recoder emits no digits for them and appends the index to a compact list (one atomic per wave); k_msm_ones
// then sums the listed bases directly.  Without this they would all land in bucket 1 of window 0 -- one giant bucket that
// serialises the fine sort of its sub-group and the partial merge.
__device__ __forceinline__ bool zl_take_one(const uint32_t* s, uint32_t i, uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count) {
    const bool one = s[0] == 1u && (s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7]) == 0u;
    const uint64_t m = __ballot(one);
    if (m == 0) return false;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)__ffsll((long long)m) - 1u) base = atomicAdd(ones_count, (uint32_t)__popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (one) ones_list[base + __popcll(m & ((1ull << lane) - 1ull))] = i;
    return one;
}

// The ABI takes canonical scalars (< r, what ark's into_repr() yields).  A scalar with bits at or above SC_BITS cannot be one; the window
// layout (W = ceil((SC_BITS + 1) / c) windows, spread top window) silently drops or misplaces such bits, so the recoder flags them and the
// call returns ZL_EINVAL instead of a wrong sum.
__device__ __forceinline__ void zl_flag_wide_scalar(uint32_t top_word, int sc_bits, uint32_t* __restrict__ bad) {
    if (bad && (top_word >> (sc_bits - 224)) != 0u) atomicOr(bad, 1u);
}
// GLV half-scalars (k_glv_split): a 127-bit magnitude in words 0..3 and the sign in bit 31 of word 7.  The recoders strip the sign off
// the record and fold it into the sign of every digit.
__device__ __forceinline__ uint32_t zl_take_sign(uint32_t& top_word, int glv) {
    if (!glv) return 0u;
    const uint32_t sg = top_word >> 31;
    top_word &= 0x7FFFFFFFu;
    return sg;
}
// MODE 0: histogram; MODE 1: scatter (cursor initialised with the bucket offsets)
template <int MODE>
__global__ void __launch_bounds__(256) k_msm_digits(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W,
                                                    uint32_t* __restrict__ counters, uint32_t* __restrict__ entries,
                                                    uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                    int sc_bits, uint32_t* __restrict__ bad) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n && !(inf && inf[i]);
    const uint32_t* s = scalars + (size_t)(live ? i : 0) * 8;
    uint32_t sv[8];
    for (int k = 0; k < 8; k++) sv[k] = live ? s[k] : 0u;
    if (MODE == 0) zl_flag_wide_scalar(sv[7], sc_bits, bad);  // (this path never runs on GLV half-scalars: MsmJob::plan)
    if (MODE == 0) {
        if (zl_take_one(sv, i, ones_list, ones_count)) return;
    } else if (sv[0] == 1u && (sv[1] | sv[2] | sv[3] | sv[4] | sv[5] | sv[6] | sv[7]) == 0u) {
        return;
    }
    if (!live) return;
    const uint32_t H = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        uint32_t d = zl_get_bits(s, w * c, c) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > H) { d = (2 * H) - d; neg = 1; carry = 1; }  // d - 2^c < 0, magnitude 2^c - d in [0, H-1]
        if (d == 0) continue;
        uint32_t bucket = (uint32_t)w * H + (d - 1);
        if (MODE == 0) {
            atomicAdd(&counters[bucket], 1u);
        } else {
            uint32_t pos = atomicAdd(&counters[bucket], 1u);
            entries[pos] = i | (neg << 31);
        }
    }
}

// ---- LDS counting sort (c <= 16): no global atomics -------------------------------------------------------------
// k_msm_recode: one lane per scalar, all W signed digits, coalesced u16 stores digits[w][i]:
//   0xFFFF = zero digit, else (neg << 15) | (magnitude - 1)      (negative magnitudes are <= H-1, so 0xFFFF is free)
static __global__ void __launch_bounds__(256) k_msm_recode(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, int spread_t, int glv, uint16_t* __restrict__ digits,
                                                             uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                             int sc_bits, uint32_t* __restrict__ bad) {
    ZL_SIDE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)(live ? i : 0) * 8);
    uint4 lo = sp[0], hi = sp[1];
    if (live) zl_flag_wide_scalar(hi.w, sc_bits, bad);
    if (!live || (inf && inf[i])) lo = hi = make_uint4(0, 0, 0, 0);  // a base at infinity contributes nothing: its scalar is dropped here
    uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (zl_take_one(s, i, ones_list, ones_count)) s[0] = 0;  // listed: contributes no digits (a negative half-scalar has its sign bit set: never listed)
    if (!live) return;
    const uint32_t sg = zl_take_sign(s[7], glv);
    const uint32_t H = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        const int pos = w * c;
        const int word = pos >> 5, sh = pos & 31;
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {  // register-resident select instead of a dynamically indexed array
            if (k == word) v |= s[k];
            if (k == word + 1) v |= (uint64_t)s[k] << 32;
        }
        uint32_t d = ((uint32_t)(v >> sh) & ((1u << c) - 1)) + carry;
        uint32_t neg = 0;
        carry = 0;
        // tie d == H: +H for a positive scalar, -H (carry 1) for a negative half-scalar, whose digits are all flipped below -- either way the
        // magnitude H only ever appears with a clear sign bit in the code, so 0xFFFF stays free at c = 16 too.  (The tie cannot happen in the top
        // window of a GLV half-scalar, where the carry would be lost: |k_i| <= lambda / 2 + 1 < 0.68 * 2^127.)
        if (d > H - sg) { d = 2 * H - d; neg = 1; carry = 1; }
        uint32_t b = d - 1;
        // a narrow top window (spread_t + 1 bits) is spread over its whole bucket set like in k_msm_recode_wide; its digits are never
        // negative (magnitudes <= 2^spread_t <= H), so the code 0xFFFF stays free
        if (spread_t >= 0 && w == W - 1) b |= (i & ((1u << (c - 1 - spread_t)) - 1u)) << spread_t;
        // sign of a GLV half-scalar: folded into the digit's sign.  A flipped digit of magnitude H (bucket H - 1, sign set) would be the code
        // 0xFFFF = "zero digit" when c = 16; MsmJob::plan sends GLV jobs with c = 16 through the wide sort, whose zero marker lives in hi8
        digits[(size_t)w * n + i] = d == 0 ? (uint16_t)0xFFFF : (uint16_t)(((neg ^ sg) << 15) | b);
    }
}
// block (slice, w): private LDS histogram of window w over a slice of the scalars -> counts[slice][w*H + bin]
static __global__ void __launch_bounds__(1024) k_msm_hist_lds(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t per_slice, uint32_t NB,
                                                        uint32_t* __restrict__ counts) {
    ZL_SIDE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const uint32_t slice = blockIdx.x, w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < H; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    const uint32_t lo = slice * per_slice, hi = min(n, lo + per_slice);
    const uint16_t* dw = digits + (size_t)w * n;
    {
        // 16-byte loads over the aligned middle of the slice (8 digits per lane and load), scalar loads at its ragged ends
        const size_t row0 = (size_t)w * n;
        uint32_t a = lo, b = hi;
        while (a < b && ((row0 + a) & 7)) a++;
        b = a + ((b - a) & ~7u);
        for (uint32_t i = lo + threadIdx.x; i < a; i += blockDim.x) {
            const uint32_t code = dw[i];
            if (code != 0xFFFFu) atomicAdd(&hist[code & 0x7FFFu], 1u);
        }
        const uint4* dv = reinterpret_cast<const uint4*>(dw + a);
        for (uint32_t j = threadIdx.x; j < (b - a) / 8; j += blockDim.x) {
            const uint4 v = dv[j];
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t code = (words[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                if (code != 0xFFFFu) atomicAdd(&hist[code & 0x7FFFu], 1u);
            }
        }
        for (uint32_t i = b + threadIdx.x; i < hi; i += blockDim.x) {
            const uint32_t code = dw[i];
            if (code != 0xFFFFu) atomicAdd(&hist[code & 0x7FFFu], 1u);
        }
    }
    __syncthreads();
    uint32_t* out = counts + (size_t)slice * NB + (size_t)w * H;
    for (uint32_t b = threadIdx.x; b < H; b += blockDim.x) out[b] = hist[b];
}
// lane per bucket: counts[slice][bucket] -> exclusive prefix over slices (in place), tot[bucket] = sum
static __global__ void __launch_bounds__(256) k_msm_slice_prefix(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ tot) {
    ZL_SIDE_PRIO();
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    uint32_t run = 0;
    for (uint32_t sl = 0; sl < nslices; sl++) {
        const uint32_t v = counts[(size_t)sl * NB + b];
        counts[(size_t)sl * NB + b] = run;
        run += v;
    }
    tot[b] = run;
}
// block (range, w): owns buckets [range*RB, (range+1)*RB) of window w, streams ALL digits of the window (coalesced u16) and
// scatters the matching entries through LDS cursors.  One block writes one contiguous, L2-resident slice of the entry list,
// so partial-sector writes merge in L2 (the slice-owned variant measured 8.5 GB of HBM writes for 1 GB of entries).
static __global__ void __launch_bounds__(1024) k_msm_scatter_range(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t RB,
                                                                     const uint32_t* __restrict__ offsets, uint32_t* __restrict__ entries,
                                                                     const uint32_t* __restrict__ slice_prefix, uint32_t NB, uint32_t nslices, uint32_t per_slice,
                                                                     uint32_t parts, uint32_t W) {
    ZL_SIDE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);
    // XCD-aware order (workgroup L runs on XCD L % 8, each XCD has its own L2): all blocks that stream the digit row of one window sit on ONE
    // XCD, so the row crosses the fabric once instead of once per XCD that hosts one of its `ranges x parts` readers
    const uint32_t ranges = gridDim.y, xcd = blockIdx.x & 7u, jj = blockIdx.x >> 3;
    const uint32_t w = xcd + 8u * jj, range = blockIdx.y, part = blockIdx.z;
    if (w >= W) return;
    (void)ranges;
    const uint32_t b0 = range * RB;
    // part p of the digit row = slices [p nslices / parts, (p + 1) nslices / parts) of k_msm_hist_lds: its cursors start behind the entries of
    // the earlier slices (slice_prefix[slice][bucket] = exclusive prefix over slices, k_msm_slice_prefix / k_msm_prefix_small).
    const uint32_t s0 = (uint32_t)((uint64_t)part * nslices / parts), s1 = (uint32_t)((uint64_t)(part + 1) * nslices / parts);
    const uint32_t lo = min(n, s0 * per_slice), hi = part + 1 == parts ? n : min(n, s1 * per_slice);
    const uint32_t* os = offsets + (size_t)w * H + b0;
    const uint32_t* sp = slice_prefix + (size_t)s0 * NB + (size_t)w * H + b0;
    for (uint32_t b = threadIdx.x; b < RB; b += blockDim.x) cur[b] = (b0 + b < H) ? os[b] + (parts > 1 ? sp[b] : 0u) : 0;
    __syncthreads();
    const uint16_t* dw = digits + (size_t)w * n;
    auto take1 = [&](uint32_t i) {
        const uint32_t code = dw[i];
        const uint32_t bucket = code & 0x7FFFu;
        if (code != 0xFFFFu && bucket - b0 < RB) {
            const uint32_t pos = atomicAdd(&cur[bucket - b0], 1u);
            entries[pos] = i | ((code >> 15) << 31);
        }
    };
    // all eight LDS cursor atomics of a 16-byte load are issued before the first store needs its position
    // A digit matches this block's bucket range with probability 1 / ranges, so eight predicated (atomic, store) pairs per load would each run
    // with a few lanes: the hits of a lane's eight digits are collected in a bit mask and the wave loops max-over-lanes(hits) times (~3) instead.
    auto take8 = [&](const uint4& v, uint32_t i0) {
        uint32_t mask = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t word = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
            const uint32_t code = (word >> ((k & 1) * 16)) & 0xFFFFu;
            if (code != 0xFFFFu && (code & 0x7FFFu) - b0 < RB) mask |= 1u << k;
        }
        while (mask) {
            const uint32_t k = (uint32_t)__builtin_ctz(mask);
            mask &= mask - 1u;
            const uint32_t word = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
            const uint32_t code = (word >> ((k & 1u) * 16u)) & 0xFFFFu;
            const uint32_t pos = atomicAdd(&cur[(code & 0x7FFFu) - b0], 1u);
            entries[pos] = (i0 + k) | ((code >> 15) << 31);
        }
    };
    // 16-byte loads (8 digits) over the aligned middle of [lo, hi), scalar loads at its ragged ends
    const size_t row0 = (size_t)w * n;
    uint32_t a = lo, e = hi;
    while (a < e && ((row0 + a) & 7)) a++;
    e = a + ((e - a) & ~7u);
    for (uint32_t i = lo + threadIdx.x; i < a; i += blockDim.x) take1(i);
    const uint4* dv = reinterpret_cast<const uint4*>(dw + a);
    const uint32_t cnt8 = (e - a) / 8, stride = blockDim.x;
    uint32_t j = threadIdx.x;
    for (; j + stride < cnt8; j += 2 * stride) {
        const uint4 v0 = dv[j], v1 = dv[j + stride];
        take8(v0, a + j * 8);
        take8(v1, a + (j + stride) * 8);
    }
    for (; j < cnt8; j += stride) take8(dv[j], a + j * 8);
    for (uint32_t i = e + threadIdx.x; i < hi; i += blockDim.x) take1(i);
}

// ---- wide windows over precomputed multiples (zl_bases_precompute): ONE bucket set of 2^(c-1) buckets, c up to 24 ----------
// Every (scalar i, window w) digit d contributes d * (2^(c w) P_i), and 2^(c w) P_i is a table entry, so all windows
// share the buckets: n*W mixed adds into 2^(c-1) buckets and a single bucket reduction.  The bucket index has up to 23
// bits, so the counting sort is two-level: partition by the high bits (group = bucket >> 15), then the LDS sort per group.
// spread_t >= 0 (plain wide windows only): the top window holds just spread_t + 1 bits, so its 2^spread_t magnitudes would crowd n entries
// into 2^spread_t buckets (one sort group) while its bucket set has 2^(c-1).  It is spread over the whole set instead: bucket =
// (low bits of the point index) << spread_t | (magnitude - 1); the reduction weights those buckets by their low spread_t bits only.
static __global__ void __launch_bounds__(256) k_msm_recode_wide(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, uint32_t gw, int spread_t, int glv,
                                                                  uint16_t* __restrict__ lo16, uint8_t* __restrict__ hi8,
                                                                  uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                                  int sc_bits, uint32_t* __restrict__ bad) {
    ZL_SIDE_PRIO();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)(live ? i : 0) * 8);
    uint4 lo = sp[0], hi = sp[1];
    if (live) zl_flag_wide_scalar(hi.w, sc_bits, bad);
    if (!live || (inf && inf[i])) lo = hi = make_uint4(0, 0, 0, 0);  // a base at infinity contributes nothing
    uint32_t s[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (zl_take_one(s, i, ones_list, ones_count)) s[0] = 0;
    if (!live) return;
    const uint32_t sg = zl_take_sign(s[7], glv);
    const uint32_t H = 1u << (c - 1);
    uint32_t carry = 0;
    for (int w = 0; w < W; w++) {
        const int pos = w * c;
        const int word = pos >> 5, sh = pos & 31;
        uint64_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (k == word) v |= s[k];
            if (k == word + 1) v |= (uint64_t)s[k] << 32;
        }
        uint32_t d = ((uint32_t)(v >> sh) & ((1u << c) - 1)) + carry;
        uint32_t neg = 0;
        carry = 0;
        if (d > H) { d = 2 * H - d; neg = 1; carry = 1; }
        uint32_t b = d - 1;  // bucket (d != 0)
        if (spread_t >= 0 && w == W - 1) b |= (i & ((1u << (c - 1 - spread_t)) - 1u)) << spread_t;
        lo16[(size_t)w * n + i] = (uint16_t)((b & 0x7FFFu) | ((neg ^ sg) << 15));
        hi8[(size_t)w * n + i] = d == 0 ? (uint8_t)0xFF : (uint8_t)((uint32_t)w * gw + (b >> 15));  // gw = groups per window (0: merged set)
    }
}
// block (slice, w): histogram of the group ids of window w over a slice of scalars -> counts[(g*W + w)*nslices + slice]
static __global__ void __launch_bounds__(256) k_msm_part_hist(const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W, uint32_t G, uint32_t per_slice,
                                                                uint32_t nslices, uint32_t* __restrict__ counts) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t hist[256];
    const uint32_t slice = blockIdx.x, w = blockIdx.y;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = slice * per_slice, hi = min(n, lo + per_slice);
    const uint8_t* hw = hi8 + (size_t)w * n;
    // 16 group ids per 16-B load over the aligned body (row base w*n and lo are multiples of 16 for the usual power-of-two n)
    const bool al = ((((size_t)w * n) | lo) & 15) == 0;
    const uint32_t body1 = al ? lo + ((hi - lo) & ~15u) : lo;
    const uint4* hv = reinterpret_cast<const uint4*>(hw);
    for (uint32_t i16 = lo / 16 + threadIdx.x; i16 < body1 / 16; i16 += blockDim.x) {
        const uint4 v = hv[i16];
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t g = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
            if (g != 0xFFu) atomicAdd(&hist[g], 1u);
        }
    }
    for (uint32_t i = body1 + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t g = hw[i];
        if (g != 0xFFu) atomicAdd(&hist[g], 1u);
    }
    __syncthreads();
    if (threadIdx.x < G) counts[((size_t)threadIdx.x * W + w) * nslices + slice] = hist[threadIdx.x];
}
// group range [s, e) from the scanned partition counters
__device__ __forceinline__ void zl_group_range(const uint32_t* __restrict__ part_off, uint32_t g, uint32_t G, uint32_t stride, uint32_t E, uint32_t& s,
                                               uint32_t& e) {
    s = part_off[(size_t)g * stride];
    e = (g + 1 < G) ? part_off[(size_t)(g + 1) * stride] : E;
}
// ---- levels 2 and 3 of the wide sort: 128 sub-groups of 256 buckets per group, then an LDS-staged bucket sort -----------------
// Measured on gfx950: a scattered 4-byte store costs about one 64-B L2 write transaction (~25 ps each at 2^24*12 entries), while
// a partition into <= 128 streams writes long runs and an LDS-staged sort writes fully coalesced.  So: group (32768 buckets) ->
// sub-group (256 buckets, ~25k entries) by one more partition pass, then one block per sub-group sorts its entries by bucket in
// LDS and copies them out linearly.
// block (slice, g): histogram of the sub-group id ((lo >> 8) & 127) over a slice of group g -> counts[(g*128 + sub)*fslices + slice]
static __global__ void __launch_bounds__(256) k_msm_sub_hist(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_off, uint32_t G,
                                                               uint32_t stride, const uint32_t* __restrict__ total, uint32_t fslices,
                                                               uint32_t* __restrict__ counts) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t hist[128];
    const uint32_t slice = blockIdx.x, g = blockIdx.y;
    if (threadIdx.x < 128) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t s, e;
    zl_group_range(part_off, g, G, stride, *total, s, e);
    const uint32_t per = (e - s + fslices - 1) / fslices;
    const uint32_t lo = min(e, s + slice * per), hi = min(e, lo + per);
    // 16-B loads (8 codes per lane) over the aligned body, scalar head / tail: the kernel is latency bound otherwise
    const uint32_t body0 = min(hi, (lo + 7u) & ~7u), body1 = max(body0, hi & ~7u);
    for (uint32_t j = lo + threadIdx.x; j < body0; j += blockDim.x) atomicAdd(&hist[(part_lo[j] >> 8) & 127u], 1u);
    const uint4* dv = reinterpret_cast<const uint4*>(part_lo);
    for (uint32_t j8 = body0 / 8 + threadIdx.x; j8 < body1 / 8; j8 += blockDim.x) {
        const uint4 v = dv[j8];
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; k++) atomicAdd(&hist[((words[k >> 1] >> ((k & 1) * 16)) >> 8) & 127u], 1u);
    }
    for (uint32_t j = body1 + threadIdx.x; j < hi; j += blockDim.x) atomicAdd(&hist[(part_lo[j] >> 8) & 127u], 1u);
    __syncthreads();
    if (threadIdx.x < 128) counts[((size_t)g * 128 + threadIdx.x) * fslices + slice] = hist[threadIdx.x];
}
// ---- LDS-staged partition (levels 1 and 2) ----------------------------------------------------------------------------------------
// A direct multi-stream scatter issues, per store instruction, up to 64 four-byte writes into different cache lines.  Staging a tile
// of ZL_PT entries in LDS first (histogram -> bin starts -> scatter inside LDS) turns the global writes into runs of
// ~ZL_PT/bins consecutive entries per bin, written by consecutive lanes.  BINS <= 256.  Entries of a tile are held in registers
// between the histogram and the LDS scatter (ZL_PT / 256 per lane).
#define ZL_PT 4096
struct PartStage {
    uint32_t gcur[256];   // global cursor of every bin (this block's private stream)
    uint32_t hist[256];
    uint32_t start[257];
    uint32_t idx[ZL_PT];
    uint16_t code[ZL_PT];
};
// after hist[] is final for the tile: exclusive scan (block of 256 lanes) -> start[]
__device__ __forceinline__ void zl_part_scan(PartStage& st) {
    // 256 values, 4 waves: wave-level inclusive scan with shuffles, then wave offsets through LDS
    __shared__ uint32_t wsum[4];
    const uint32_t v = st.hist[threadIdx.x];
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if ((threadIdx.x & 63) >= (uint32_t)off) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) base += wsum[w];
    st.start[threadIdx.x] = base + x - v;
    if (threadIdx.x == 255) st.start[256] = base + x;
    __syncthreads();
}
// copy the staged tile out: staged position j belongs to bin b(j); destination = gcur[b] + (j - start[b]).  BINFN maps the staged
// 16-bit code to its bin, OUTFN to the code stored at the next level.
template <class BinFn, class OutFn>
__device__ __forceinline__ void zl_part_flush(PartStage& st, uint32_t cnt, uint16_t* __restrict__ out_lo, uint32_t* __restrict__ out_idx, BinFn binfn, OutFn outfn) {
    for (uint32_t j = threadIdx.x; j < cnt; j += 256) {
        const uint32_t code = st.code[j];
        const uint32_t b = binfn(code);
        const uint32_t dest = st.gcur[b] + (j - st.start[b]);
        out_lo[dest] = (uint16_t)outfn(code);
        out_idx[dest] = st.idx[j];
    }
    __syncthreads();
    st.gcur[threadIdx.x] += st.hist[threadIdx.x];
    st.hist[threadIdx.x] = 0;
    __syncthreads();
}
// level 1, block (slice, w): (group id from hi8, code from lo16) -> group-partitioned lists.  The staged 16-bit code cannot carry the
// 8-bit group id too, so the group rides in a parallel LDS byte array.
static __global__ void __launch_bounds__(256) k_msm_part_scatter_st(const uint16_t* __restrict__ lo16, const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W,
                                                                      uint32_t G, uint32_t per_slice, uint32_t nslices, const uint32_t* __restrict__ part_off,
                                                                      uint32_t table_stride, uint32_t first, uint16_t* __restrict__ out_lo,
                                                                      uint32_t* __restrict__ out_idx) {
    ZL_SIDE_PRIO();
    __shared__ PartStage st;
    __shared__ uint8_t grp[ZL_PT];
    const uint32_t slice = blockIdx.x, w = blockIdx.y;
    st.gcur[threadIdx.x] = threadIdx.x < G ? part_off[((size_t)threadIdx.x * W + w) * nslices + slice] : 0u;
    st.hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = slice * per_slice, hi = min(n, lo + per_slice);
    const uint8_t* hw = hi8 + (size_t)w * n;
    const uint16_t* lw = lo16 + (size_t)w * n;
    const uint32_t add = w * table_stride + first;
    const bool al = ((((size_t)w * n) | lo) & 15) == 0;  // row base and slice start 16-aligned: 16 entries per 16-B load of hi8
    constexpr int EPT = ZL_PT / 256;                     // 16 entries per lane per tile
    for (uint32_t t0 = lo; t0 < hi; t0 += ZL_PT) {
        const uint32_t j0 = t0 + threadIdx.x * EPT;      // this lane's 16 consecutive entries
        uint32_t g[EPT], code[EPT], rank[EPT];
        if (al && j0 + EPT <= hi) {
            const uint4 v = *reinterpret_cast<const uint4*>(hw + j0);
            const uint4 l0 = *reinterpret_cast<const uint4*>(lw + j0), l1 = *reinterpret_cast<const uint4*>(lw + j0 + 8);
            const uint32_t words[4] = {v.x, v.y, v.z, v.w};
            const uint32_t lws[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                g[k] = (words[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                code[k] = (lws[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
            }
        } else {
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const bool ok = j0 + k < hi;
                g[k] = ok ? hw[j0 + k] : 0xFFu;
                code[k] = ok ? lw[j0 + k] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; k++) rank[k] = g[k] != 0xFFu ? atomicAdd(&st.hist[g[k]], 1u) : 0u;
        __syncthreads();
        zl_part_scan(st);
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            if (g[k] != 0xFFu) {
                const uint32_t pos = st.start[g[k]] + rank[k];
                st.code[pos] = (uint16_t)code[k];
                st.idx[pos] = add + j0 + k;
                grp[pos] = (uint8_t)g[k];
            }
        }
        __syncthreads();
        const uint32_t cnt = st.start[256];
        for (uint32_t j = threadIdx.x; j < cnt; j += 256) {
            const uint32_t b = grp[j];
            const uint32_t dest = st.gcur[b] + (j - st.start[b]);
            out_lo[dest] = st.code[j];
            out_idx[dest] = st.idx[j];
        }
        __syncthreads();
        st.gcur[threadIdx.x] += st.hist[threadIdx.x];
        st.hist[threadIdx.x] = 0;
        __syncthreads();
    }
}
// level 2, block (slice, g): sub-group id = (code >> 8) & 127 -> sub-group-partitioned lists (fine bucket + sign kept in the code)
static __global__ void __launch_bounds__(256) k_msm_sub_scatter_st(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_idx,
                                                                     const uint32_t* __restrict__ part_off, uint32_t G, uint32_t stride,
                                                                     const uint32_t* __restrict__ total, uint32_t fslices, const uint32_t* __restrict__ sub_off,
                                                                     uint16_t* __restrict__ out_lo, uint32_t* __restrict__ out_idx) {
    ZL_SIDE_PRIO();
    __shared__ PartStage st;
    const uint32_t slice = blockIdx.x, g = blockIdx.y;
    st.gcur[threadIdx.x] = threadIdx.x < 128 ? sub_off[((size_t)g * 128 + threadIdx.x) * fslices + slice] : 0u;
    st.hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t s, e;
    zl_group_range(part_off, g, G, stride, *total, s, e);
    const uint32_t per = (e - s + fslices - 1) / fslices;
    const uint32_t lo = min(e, s + slice * per), hi = min(e, lo + per);
    constexpr int EPT = ZL_PT / 256;  // 16 entries per lane per tile, as two 8-entry groups aligned to 8 (16-B loads of the u16 codes)
    const uint32_t a0 = lo & ~7u;     // tiles start on an 8-aligned index; entries outside [lo, hi) are masked
    for (uint32_t t0 = a0; t0 < hi; t0 += ZL_PT) {
        const uint32_t j0 = t0 + threadIdx.x * EPT;
        uint32_t code[EPT], idx[EPT], rank[EPT];
        bool ok[EPT];
        if (j0 >= lo && j0 + EPT <= hi) {
            const uint4 c0 = *reinterpret_cast<const uint4*>(part_lo + j0), c1 = *reinterpret_cast<const uint4*>(part_lo + j0 + 8);
            const uint4* iv = reinterpret_cast<const uint4*>(part_idx + j0);
            const uint4 i0 = iv[0], i1 = iv[1], i2 = iv[2], i3 = iv[3];
            const uint32_t cw[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const uint32_t iw[16] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w, i3.x, i3.y, i3.z, i3.w};
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                code[k] = (cw[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
                idx[k] = iw[k];
                ok[k] = true;
            }
        } else {
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                ok[k] = j0 + k >= lo && j0 + k < hi;
                code[k] = ok[k] ? part_lo[j0 + k] : 0u;
                idx[k] = ok[k] ? part_idx[j0 + k] : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < EPT; k++) rank[k] = ok[k] ? atomicAdd(&st.hist[(code[k] >> 8) & 127u], 1u) : 0u;
        __syncthreads();
        zl_part_scan(st);
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            if (ok[k]) {
                const uint32_t pos = st.start[(code[k] >> 8) & 127u] + rank[k];
                st.code[pos] = (uint16_t)code[k];
                st.idx[pos] = idx[k];
            }
        }
        __syncthreads();
        zl_part_flush(st, st.start[256], out_lo, out_idx, [](uint32_t c) { return (c >> 8) & 127u; },
                      [](uint32_t c) { return (c & 0xFFu) | (c & 0x8000u); });
    }
}
// block per sub-group: histogram of its 256 buckets -> counts[sg*256 + bin]  (sg*256 + bin IS the bucket index)
static __global__ void __launch_bounds__(256) k_msm_fine_hist(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ sub_off, uint32_t SG,
                                                                uint32_t fslices, const uint32_t* __restrict__ total, uint32_t* __restrict__ counts) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t hist[256];
    const uint32_t sg = blockIdx.x;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t s = sub_off[(size_t)sg * fslices], e = (sg + 1 < SG) ? sub_off[(size_t)(sg + 1) * fslices] : *total;
    const uint32_t body0 = min(e, (s + 7u) & ~7u), body1 = max(body0, e & ~7u);
    for (uint32_t j = s + threadIdx.x; j < body0; j += blockDim.x) atomicAdd(&hist[lo2[j] & 0xFFu], 1u);
    const uint4* dv = reinterpret_cast<const uint4*>(lo2);
    for (uint32_t j8 = body0 / 8 + threadIdx.x; j8 < body1 / 8; j8 += blockDim.x) {
        const uint4 v = dv[j8];
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; k++) atomicAdd(&hist[(words[k >> 1] >> ((k & 1) * 16)) & 0xFFu], 1u);
    }
    for (uint32_t j = body1 + threadIdx.x; j < e; j += blockDim.x) atomicAdd(&hist[lo2[j] & 0xFFu], 1u);
    __syncthreads();
    counts[(size_t)sg * 256 + threadIdx.x] = hist[threadIdx.x];
}
// block per sub-group: sort the sub-group's entries by bucket inside LDS (cursors = bucket offsets relative to the sub-group),
// then copy the staged run to the entry list with consecutive lanes writing consecutive words.  Oversized sub-groups (the narrow top
// window concentrates its entries in few buckets; skewed scalars) are cut into tiles of ZL_BT entries and queued for
// k_msm_fine_sort_big: (sub-group, first tile, tiles) records, index and tile base reserved with ONE 64-bit atomic so that the
// record order is the tile order.
#define ZL_BT 16384
static __global__ void __launch_bounds__(1024) k_msm_fine_sort(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
                                                                const uint32_t* __restrict__ sub_off, uint32_t SG, uint32_t fslices,
                                                                const uint32_t* __restrict__ total, const uint32_t* __restrict__ offsets, uint32_t cap,
                                                                uint32_t* __restrict__ entries, unsigned long long* __restrict__ big_head,
                                                                uint32_t* __restrict__ big_items) {
    ZL_SIDE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* cur = reinterpret_cast<uint32_t*>(smem);  // [256]
    uint32_t* stage = cur + 256;                        // [cap]
    const uint32_t sg = blockIdx.x;
    const uint32_t s = sub_off[(size_t)sg * fslices], e = (sg + 1 < SG) ? sub_off[(size_t)(sg + 1) * fslices] : *total;
    const uint32_t base = offsets[(size_t)sg * 256];  // first entry slot of this sub-group (= s: same count, same order of groups)
    const uint32_t len = e - s;
    if (len > cap) {
        if (threadIdx.x == 0) {
            const uint32_t tiles = (len + ZL_BT - 1) / ZL_BT;
            const unsigned long long old = atomicAdd(big_head, (1ull << 32) | tiles);
            const uint32_t item = (uint32_t)(old >> 32);
            big_items[2 * item] = sg;
            big_items[2 * item + 1] = (uint32_t)old;  // first tile
        }
        return;
    }
    for (uint32_t b = threadIdx.x; b < 256; b += blockDim.x) cur[b] = offsets[(size_t)sg * 256 + b] - base;
    __syncthreads();
    for (uint32_t j = s + threadIdx.x; j < e; j += blockDim.x) {
        const uint32_t code = lo2[j];
        const uint32_t pos = atomicAdd(&cur[code & 0xFFu], 1u);
        stage[pos] = idx2[j] | ((code >> 15) << 31);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) entries[base + k] = stage[k];
}
// tiles of the oversized sub-groups, any number of blocks: a tile is sorted by bucket inside LDS, every bucket's run reserves its
// place in the entry list with one atomic on the bucket's global cursor (initialised to the bucket offsets), and the runs are copied
// out by consecutive lanes.  Which tile lands first inside a bucket is not deterministic; a bucket's SUM does not depend on the
// order of its entries (group law), so results are unchanged.
static __global__ void __launch_bounds__(1024) k_msm_fine_sort_big(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
                                                                    const uint32_t* __restrict__ sub_off, uint32_t SG, uint32_t fslices,
                                                                    const uint32_t* __restrict__ total, const unsigned long long* __restrict__ big_head,
                                                                    const uint32_t* __restrict__ big_items, uint32_t* __restrict__ cursor,
                                                                    uint32_t* __restrict__ entries) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t hist[256], start[257], gbase[256], wsum[4], item_sg, item_tile0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem);  // [ZL_BT]
    const unsigned long long head = *big_head;
    const uint32_t items = (uint32_t)(head >> 32), tiles = (uint32_t)head;
    constexpr int EPT = ZL_BT / 1024;
    for (uint32_t g = blockIdx.x; g < tiles; g += gridDim.x) {
        if (threadIdx.x == 0) {  // record with the largest first-tile <= g (records are in tile order)
            uint32_t lo = 0, hi = items;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (big_items[2 * mid + 1] <= g) lo = mid; else hi = mid;
            }
            item_sg = big_items[2 * lo];
            item_tile0 = big_items[2 * lo + 1];
        }
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t sg = item_sg;
        const uint32_t s = sub_off[(size_t)sg * fslices], e = (sg + 1 < SG) ? sub_off[(size_t)(sg + 1) * fslices] : *total;
        const uint32_t t0 = s + (g - item_tile0) * ZL_BT, t1 = min(e, t0 + ZL_BT), cnt = t1 - t0;
        uint32_t val[EPT], bin[EPT], rank[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const uint32_t j = t0 + k * 1024 + threadIdx.x;
            bin[k] = 0xFFFFFFFFu;
            if (j < t1) {
                const uint32_t code = lo2[j];
                val[k] = idx2[j] | ((code >> 15) << 31);
                bin[k] = code & 0xFFu;
                rank[k] = atomicAdd(&hist[bin[k]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) {  // exclusive scan of the 256 counts (4 waves) + run reservation
            const uint32_t v = hist[threadIdx.x];
            uint32_t x = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t y = __shfl_up(x, off);
                if ((threadIdx.x & 63) >= (uint32_t)off) x += y;
            }
            if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
            gbase[threadIdx.x] = v ? atomicAdd(&cursor[(size_t)sg * 256 + threadIdx.x], v) : 0u;
            hist[threadIdx.x] = x - v;  // exclusive inside the wave; wave bases are added below
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t wb = 0;
            for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) wb += wsum[w];
            start[threadIdx.x] = hist[threadIdx.x] + wb;
            if (threadIdx.x == 255) start[256] = cnt;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; k++)
            if (bin[k] != 0xFFFFFFFFu) stage[start[bin[k]] + rank[k]] = val[k];
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < cnt; j += 1024) {
            uint32_t lo = 0, hi = 256;  // bin with start[bin] <= j < start[bin + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (start[mid] <= j) lo = mid; else hi = mid;
            }
            entries[gbase[lo] + (j - start[lo])] = stage[j];
        }
        __syncthreads();
    }
}

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

// ------------------------------------------------------------------------------------------------ scan
// exclusive scan of `count` u32 values, 3 launches; out[count] = total
#define SCAN_ITEMS 16
#define SCAN_BLOCK 256
static __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ block_sums) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < count) s += in[base + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_BLOCK / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0];
}
static __global__ void __launch_bounds__(1024) k_scan_top(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t* __restrict__ total_out,
                                                          const uint32_t* __restrict__ flag_in = nullptr /* copied to total_out[1] */) {
    ZL_SIDE_PRIO();
    // single block: exclusive scan of block_sums in place
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nblocks; base += 1024) {
        uint32_t idx = base + threadIdx.x;
        uint32_t v = idx < nblocks ? block_sums[idx] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t incl = sh[threadIdx.x];
        if (idx < nblocks) block_sums[idx] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (flag_in) total_out[1] = *flag_in;
    }
}
// Small bucket counts (NB <= 16384): slice prefix + the three scan launches as ONE block of 1024 lanes, 16 consecutive buckets per lane.
// counts[slice][bucket] -> in-place exclusive prefix over slices; offsets[b] = cursor[b] = exclusive prefix over buckets; offsets[NB] = total,
// offsets[NB + 1] = *flag_in (as k_scan_top).  Four dependent launches of a few microseconds each are what a small job's sort phase consists of.
static __global__ void __launch_bounds__(1024) k_msm_prefix_small(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ cursor, const uint32_t* __restrict__ flag_in) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t sh[1024];
    const uint32_t base = threadIdx.x * 16;
    uint32_t v[16];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t run = 0;
        if (base + k < NB)
            for (uint32_t sl = 0; sl < nslices; sl++) {
                const uint32_t x = counts[(size_t)sl * NB + base + k];
                counts[(size_t)sl * NB + base + k] = run;
                run += x;
            }
        v[k] = run;
        s += run;
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t t = threadIdx.x >= (uint32_t)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (base + k < NB) { offsets[base + k] = run; cursor[base + k] = run; }
        run += v[k];
    }
    if (threadIdx.x == 1023) {
        offsets[NB] = sh[1023];
        if (flag_in) offsets[NB + 1] = *flag_in;
    }
}
static __global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* __restrict__ in, uint32_t count, const uint32_t* __restrict__ block_sums,
                                                           uint32_t* __restrict__ out, uint32_t* __restrict__ out2) {
    ZL_SIDE_PRIO();
    __shared__ uint32_t sh[SCAN_BLOCK];
    uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < count) ? in[base + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_sums[blockIdx.x] + sh[threadIdx.x] - s;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (base + k < count) { out[base + k] = run; out2[base + k] = run; }
        run += v[k];
    }
}

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
                                                    const Affine<typename HotField<typename G::F>::type>* __restrict__ phib, uint32_t n_real) {
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
        const uint32_t idx = ent & 0x7fffffffu;
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
// The same chunks on a grid that does NOT fill the register file (pipelined batches): k_msm_accumulate at three waves per SIMD holds 498 of
// the 512 registers of every SIMD for as long as it runs, so the sort of the next MSM and the tail of the previous one only get onto the
// machine when it ends (rocprofv3 of a batch, profiles/r03_glv_ab.log: 3.5 ms between consecutive accumulations in which those two run alone).
// Here `wg_per_cu` workgroups of 256 lanes per CU (2: two waves per SIMD, 332 registers) loop over the chunks, which leaves a wave slot of
// ~180 registers per SIMD and all of the LDS to the side streams for the whole accumulation.  Measured and NOT used by default (see
// msm_run_jobs_t): the overlap happens, but both co-resident field-arithmetic kernels slow down far more than the gap was worth.
#define ZL_ACC_PERSIST_BLOCK 256
template <class G>
__global__ void __launch_bounds__(ZL_ACC_PERSIST_BLOCK, ZL_ACC_WAVES) k_msm_accumulate_persist(const uint32_t* __restrict__ entries, const uint32_t* __restrict__ offsets, uint32_t NB,
                                                        const Affine<typename G::F>* __restrict__ bases_,
                                                        XYZZ<typename G::F>* __restrict__ bucket_sums_,
                                                        XYZZ<typename G::F>* __restrict__ partials_, uint32_t ZL_CHUNK,
                                                        const Affine<typename G::F>* __restrict__ phib_, uint32_t n_real, uint32_t nchunks) {
    using F = typename HotField<typename G::F>::type;
    const uint32_t E = offsets[NB];
    // lanes of one wave take consecutive chunks (neighbouring entries), the grid strides over the list
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nchunks; t += gridDim.x * blockDim.x)
        zl_accumulate_chunk<G>(t, 0, E, entries, offsets, NB, reinterpret_cast<const Affine<F>*>(bases_), reinterpret_cast<XYZZ<F>*>(bucket_sums_),
                               reinterpret_cast<XYZZ<F>*>(partials_), ZL_CHUNK, reinterpret_cast<const Affine<F>*>(phib_) - n_real, n_real);
}

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

template <class G>
__global__ void __launch_bounds__(TreeLanes<G>::N) k_msm_ones(const uint32_t* __restrict__ ones_list, const uint32_t* __restrict__ ones_count,
                                                   const Affine<typename G::F>* __restrict__ bases, XYZZ<typename G::F>* __restrict__ out,
                                                   const Affine<typename G::F>* __restrict__ phib, uint32_t n_real) {
    ZL_SIDE_PRIO();
    using F = typename G::F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    XYZZ<F>* sh = reinterpret_cast<XYZZ<F>*>(smem);
    const uint32_t cnt = *ones_count;
    if (cnt == 0) {  // the usual case for uniform scalars: no block tree over 128 / 256 points at infinity (15 us of the tail of a small G2 MSM)
        if (threadIdx.x == 0) out[blockIdx.x] = XYZZ<F>::inf();
        return;
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt; j += gridDim.x * blockDim.x) {
        const uint32_t idx = ones_list[j];
        const Affine<F> P = (G::GLV && idx >= n_real) ? phib[idx - n_real] : bases[idx];  // GLV: a half-scalar k2 = 1 names phi(P)
        if (!P.is_inf()) zl::add_mixed(acc, P.x, P.y, false);
    }
    zl_block_tree<G>(sh, acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
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

// ------------------------------------------------------------------------------------------------ host driver
// developer tuning knobs (profiling sweeps only; unset in production): ZL_TUNE_CHUNK, ZL_TUNE_SEG, ZL_TUNE_FS, ZL_TUNE_RANGES (zl_tune, zl_ctx.h)
static int zl_pick_window(size_t n, int sc_bits, bool wide16 = false /* c = 16 also runs the three-level sort (GLV jobs) */) {
    // cost in accumulated entries: n per window (+10 % for c <= 16: the one-level LDS counting sort streams every window's digits once per
    // bucket range and is the slower sort at large n) + ~5.7 per bucket (merge of cut buckets, level-0 running sums, tree).  Fitted on
    // single-call times at 2^20 .. 2^24, both curves (profiles/r02_msm_sweep_plain.log, r02_msm_sweep_bn254.log): picks 16 up to 2^21,
    // 18 at 2^22 - 2^23, 19 at 2^24.  c = 17..20 run the three-level sort over W bucket sets (<= 255 sort groups).
    const double per_bucket = (double)zl_tune("ZL_TUNE_BUCKET_COST_X10", 57) / 10.0;
    double best = 1e300;
    int best_c = 2;
    for (int c = 2; c <= 20; c++) {
        int W = (sc_bits + 1 + c - 1) / c;
        if (c > 16 && (((uint64_t)W << (c - 1)) >> 15) > 255) continue;
        const bool lds_sort = c < 16 || (c == 16 && !wide16);
        double cost = (double)n * W * (lds_sort ? 1.10 : 1.0) + per_bucket * W * (double)(1u << (c - 1));
        if (cost < best) { best = cost; best_c = c; }
    }
    return best_c;
}

static int zl_pick_window_precomp(size_t n, int sc_bits) {
    // merged windows: n*W mixed adds + ONE bucket set of 2^(c-1) buckets (merge + reduce ~6 add-equivalents per bucket)
    // measured at 2^20: c = 16 and c = 20 tie for a single call (4.9 ms), 17..19 are slower (half-filled staging blocks), and inside a
    // pipeline (Groth16's five MSMs) c = 20 wins clearly: fewer additions, and the larger sort / tail are hidden
    if (n < (size_t)700000) return 16;
    if (n < ((size_t)1 << 21)) return 20;
    double best = 1e300;
    int best_c = 16;
    for (int c = 20; c <= 23; c++) {
        int W = (sc_bits + 1 + c - 1) / c;
        double cost = (double)n * W + 6.0 * (double)(1u << (c - 1));
        if (cost < best) { best = cost; best_c = c; }
    }
    return best_c;
}

// One MSM as three phases that only communicate through device buffers, so that consecutive MSMs can be pipelined on three streams
// (sort of MSM i+2 | bucket accumulation of MSM i+1 | merge / reduction tail of MSM i): plan() sizes everything, alloc() binds one of
// three buffer sets, sort() builds the bucket-sorted entry list, accumulate() is the dominant kernel, tail() leaves SETS window sums
// (+ the sum of the scalar-1 bases) in host memory, finish() does the host Horner.
template <class G>
struct MsmJob {
    using F = typename G::F;
    using X = XYZZ<F>;
    // plan
    bool pre = false;
    int c = 0, W = 0;
    int spread_t = -1;  // >= 0: the top window's entries are spread over its bucket set, weights = low spread_t bits + 1
    bool wide = false;  // three-level sort over (window, bucket) ids of up to 23 bits: table mode, or plain windows wider than 16 bits
    bool glv = false;   // the job runs on 2 n_real half-scalars of 127 bits over the points P_i and phi(P_i) (k_glv_split / k_glv_phi)
    bool phi_cached = false;  // d_phi is the handle's own copy (zl_bases::d_endo)
    bool phi_owner = false;  // this job computes the phi image of its bases in its sort phase (else it borrows d_phi from an earlier job of the call)
    int sc_bits = 0, phi_slot = -1, endo_k = 1;  // endo_k: half-scalars per scalar (2: GLV on G1, 4: GLS on BLS12-381 G2)
    int slotA = 5, slotB = 6;  // scratch slots of the sort temporaries (shared by the jobs of a pipelined batch; per buffer set when small jobs run side by side)
    size_t n_real = 0;
    uint32_t* d_vs = nullptr;            // the half-scalars (inside the sort temporaries)
    const Affine<F>* d_phi = nullptr;    // phi(P_i), i < n_real
    uint32_t H = 0, SETS = 0, NB = 0, ZL_CHUNK = 0, nchunks = 0, scan_blocks = 0, max_big = 0, max_giant = 0, Gn = 0, big_span = ZL_BIG_SPAN;
    uint32_t red_g0 = 0, red_lg0 = 0, red_blocks = 0, red_levels = 0;  // bucket reduction: block length of level 0, blocks per set, tree levels
    uint32_t roots_per_set = 0;                                         // channels of a set's root node: T, A, S_0 .. S_(levels-1)
    uint64_t maxE = 0;
    size_t n = 0, first = 0;
    const zl_bases* bsp = nullptr;
    // buffers
    uint32_t *d_counts = nullptr, *d_offsets = nullptr, *d_cursor = nullptr, *d_entries = nullptr, *d_block_sums = nullptr, *d_big_list = nullptr,
             *d_big_count = nullptr, *d_ones_count = nullptr, *d_giant_count = nullptr, *d_giant_list = nullptr, *d_ones_list = nullptr,
             *d_bigsg_items = nullptr, *d_bad_scalar = nullptr;
    unsigned long long* d_bigsg_head = nullptr;
    X *d_buckets = nullptr, *d_partials = nullptr, *d_segs = nullptr, *d_stage1 = nullptr, *d_sets = nullptr, *d_ones_parts = nullptr, *d_giant_tmp = nullptr;
    const Affine<F>* d_bases = nullptr;
    const uint8_t* d_inf = nullptr;  // per-base infinity flags of the range (null: the handle has no point at infinity)
    const uint32_t* sc = nullptr;
    // host results (pinned when pipelined)
    X* hw = nullptr;
    uint32_t* hE = nullptr;  // [0] = entries accumulated, [1] = non-canonical-scalar flag
    std::vector<X> hw_own;
    uint32_t hE_own[2] = {0, 0};

    // phi_slot_: scratch slot for the endomorphism image of the bases when this job computes it (GLV); -1 = never use the endomorphism
    int plan(zl_ctx* ctx, const zl_bases& bs, size_t first_, const void* d_scalars, size_t n_, int phi_slot_ = -1) {
        static const bool no_glv = getenv("ZL_NO_GLV") != nullptr;  // developer A/B switch
        bool try_glv = false;
        // Measured (round 3, profiles/r03_glv_ab.log): halving the bucket sets wins where the merge / reduction tails and the host Horner dominate
        // (2^16: 1.07 -> 1.00 ms, 2^18: 1.67 -> 1.63 ms, Groth16 k = 64: 3.8 -> 3.6 ms); from 2^20 on the split, the phi image of the bases
        // (read + write of every point) and the three-level sort of 2 n records cost what the tail saves (2^20: 3.73 = 3.73 ms; 2^24 single
        // call 39.2 -> 39.7 ms, pipelined 36.7 = 36.7), so large inputs keep the plain 255-bit windows.
        static const size_t glv_max = (size_t)1 << zl_tune("ZL_TUNE_GLV_MAX_LOG", 19);
        if constexpr (G::GLV) try_glv = phi_slot_ >= 0 && bs.precomp_c == 0 && !no_glv && n_ >= 1 && n_ <= glv_max && (uint64_t)n_ * G::ENDO_K < (1ull << 31);
        int rc = plan_as(ctx, bs, first_, d_scalars, n_, try_glv);
        // (c <= 3: the top window of a 127-bit half-scalar can reach magnitude H + carry; not worth a special case)
        if (!rc && glv && c <= 3) rc = plan_as(ctx, bs, first_, d_scalars, n_, false);
        // the global-atomics sort (forced plain c >= 21 beyond 255 sort groups) does not take half-scalars: plan again without them
        if (!rc && glv && !wide && c > 16) rc = plan_as(ctx, bs, first_, d_scalars, n_, false);
        phi_slot = glv ? phi_slot_ : -1;
        phi_owner = glv;
        return rc;
    }
    int plan_as(zl_ctx* ctx, const zl_bases& bs, size_t first_, const void* d_scalars, size_t n_, bool glv_) {
        glv = glv_;
        n_real = n_;
        endo_k = glv ? (int)G::ENDO_K : 1;
        n = (size_t)endo_k * n_;
        sc_bits = !glv ? (int)G::SC_BITS : (G::ENDO_K == 2 ? 127 : 64);
        first = first_;
        bsp = &bs;
        pre = bs.precomp_c > 0;  // table of 2^(c w) P_i present: all windows share one bucket set
        c = pre ? bs.precomp_c : (ctx->msm_c > 0 ? ctx->msm_c : zl_pick_window(n, sc_bits, glv && G::ENDO_K != 2));
        if (c < 2) c = 2;
        if (c > 24) c = 24;
        W = (sc_bits + 1 + c - 1) / c;
        H = 1u << (c - 1);
        SETS = pre ? 1u : (uint32_t)W;  // bucket sets
        const uint64_t NB64 = (uint64_t)SETS * H;
        maxE = (uint64_t)n * W;
        if (n >= (1ull << 31) || maxE >= (1ull << 32) || NB64 >= (1ull << 31)) return ZL_EINVAL;
        if (pre && (uint64_t)W * bs.n >= (1ull << 31)) return ZL_EINVAL;
        NB = (uint32_t)NB64;
        Gn = NB >> 15;  // sort groups of 32768 (window, bucket) ids; the group id travels in a byte, 0xFF = zero digit
        if (pre && (c < 16 || Gn < 1 || Gn > 255)) return ZL_EINVAL;
        // plain windows beyond that (c >= 21) fall back to the global-atomics sort.  GLS quarter-scalars (G2) at c = 16 take the wide sort too: their
        // narrow top window is spread over the bucket set, and a spread bucket index of all ones with the sign set would be the LDS sort's 0xFFFF =
        // "zero digit" (GLV half-scalars on G1 have a full top window and a tie rule that keeps the code free: k_msm_recode)
        wide = pre || ((c > 16 || (glv && c == 16 && G::ENDO_K != 2)) && Gn >= 1 && Gn <= 255);
        // chunk length: 64 entries per lane once there are enough entries to fill the chip (~2^18 lanes), shorter below
        // (128 once there are >= 2^20 lanes of that length: half as many cut buckets to merge; 32.8 -> 32.2 ms per pipelined 2^24 MSM)
        ZL_CHUNK = (maxE >> 7) >= (1u << 20) ? 128u : (uint32_t)ZL_CHUNK_MAX;
        while (ZL_CHUNK > 8 && maxE / ZL_CHUNK < (1u << 18)) ZL_CHUNK >>= 1;
        ZL_CHUNK = (uint32_t)std::max(8, zl_tune("ZL_TUNE_CHUNK", (int)ZL_CHUNK));
        nchunks = (uint32_t)((maxE + ZL_CHUNK - 1) / ZL_CHUNK);
        // bucket reduction (k_msm_reduce_level0 + k_msm_reduce_tree): blocks of 8 buckets (4 / 2 for smaller inputs: more lanes, shorter chains)
        {
            // measured (gpurun sweep of ZL_TUNE_SEG, round 3): 2 up to 2^16 points, 4 at 2^18 - 2^20, 8 from 2^22 on
            uint32_t g0 = NB >= (1u << 20) ? 8u : (NB >= (1u << 17) ? 4u : 2u);
            g0 = (uint32_t)std::max(2, zl_tune("ZL_TUNE_SEG", (int)g0));
            while (g0 & (g0 - 1)) g0 &= g0 - 1;
            if (g0 > H) g0 = H;
            // plain wide windows: spread the narrow top window over its whole bucket set (k_msm_recode_wide); the weight then lives in the
            // low spread_t bits of the bucket index: all of level 0's bits must be on one side of that boundary (a very narrow top
            // window, 0 < spread_t < log2 g0, shortens the level-0 blocks to 2^spread_t)
            spread_t = -1;
            if ((wide || c <= 16) && !pre) {  // (the global-atomics sort of plain c >= 21 keeps its crowded top window)
                const int top_bits = sc_bits + 1 - (W - 1) * c;  // bits of the top window incl. the carry: magnitudes <= 2^(top_bits - 1)
                if (top_bits - 1 < c - 1) {
                    spread_t = top_bits - 1;
                    if (spread_t > 0 && (1u << spread_t) < g0) g0 = 1u << spread_t;
                }
            }
            red_g0 = g0;
            red_lg0 = 31 - __builtin_clz(g0);
            red_blocks = H / g0;  // both powers of two
            red_levels = 31 - __builtin_clz(red_blocks);
            roots_per_set = red_levels + 2;
        }
        scan_blocks = (NB + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
        big_span = nchunks <= (1u << 17) ? (uint32_t)ZL_BIG_SPAN_SMALL : (uint32_t)ZL_BIG_SPAN;
        max_big = (uint32_t)(maxE / ((uint64_t)ZL_CHUNK * big_span)) + 1;
        max_giant = (uint32_t)(maxE / ((uint64_t)ZL_CHUNK * ZL_GIANT_SPAN)) + 1;
        d_bases = pre ? reinterpret_cast<const Affine<F>*>(bs.d_table) : reinterpret_cast<const Affine<F>*>(bs.d_pts) + first;
        d_inf = bs.d_inf ? reinterpret_cast<const uint8_t*>(bs.d_inf) + first : nullptr;
        sc = reinterpret_cast<const uint32_t*>(d_scalars);
        hw_own.assign((size_t)SETS * roots_per_set + 1, X::inf());
        hw = hw_own.data();
        hE = hE_own;
        return ZL_OK;
    }
    // buffer set 0, 1 or 2 (slots 0..3 + 4 / 10..13 + 19 / 14..17 + 23); the sort temporaries (slots 5, 6) are shared: the sorts of consecutive
    // jobs run in order on the sort stream
    static int phi_slot_of(int set) { return set == 3 ? 35 : 20 + set; }
    int alloc(zl_ctx* ctx, int set, bool own_sort = false) {
        void* p;
        int rc;
        const int o = set == 0 ? 0 : (set == 1 ? 10 : (set == 2 ? 14 : 28));
        if (own_sort) {  // the job's sort runs beside the other sets' sorts: its temporaries are its own, sized here (nothing is in flight yet)
            static const int A[4] = {5, 24, 26, 33}, B[4] = {6, 25, 27, 34};
            slotA = A[set];
            slotB = B[set];
            size_t a5, a6;
            sort_tmp_sizes(a5, a6);
            if (a5 && (rc = zl_scratch_get(ctx, slotA, a5, &p))) return rc;
            if (a6 && (rc = zl_scratch_get(ctx, slotB, a6, &p))) return rc;
        }
        // counters (NB+1) | offsets (NB+2: [NB] = total entries, [NB+1] = non-canonical-scalar flag) | cursor (NB+1) | block sums | big list | counts | giant list | scalar-1 list
        const size_t max_bigsg = (size_t)(maxE / 1024) + 2;  // oversized sub-groups hold > cap >= 1024 entries each
        size_t small_words = (size_t)3 * (NB + 1) + 1 + scan_blocks + 1 + max_big + max_giant + 16 + n + 2 * max_bigsg;
        if ((rc = zl_scratch_get(ctx, o + 0, small_words * 4, &p))) return rc;
        d_counts = (uint32_t*)p;
        d_offsets = d_counts + (NB + 1);
        d_cursor = d_offsets + (NB + 2);
        d_block_sums = d_cursor + (NB + 1);
        d_big_list = d_block_sums + scan_blocks + 1;
        d_big_count = d_big_list + max_big;
        d_ones_count = d_big_count + 1;
        d_giant_count = d_big_count + 2;
        d_bad_scalar = d_big_count + 3;  // zeroed with the counts; set by the recoder for a scalar with bits >= SC_BITS
        d_giant_list = d_big_count + 16;
        d_ones_list = d_giant_list + max_giant;
        d_bigsg_items = d_ones_list + n;
        d_bigsg_head = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(d_big_count + 4) + 7) & ~(uintptr_t)7);  // inside words 4..7
        if ((rc = zl_scratch_get(ctx, o + 1, maxE * 4, &p))) return rc;
        d_entries = (uint32_t*)p;
        if ((rc = zl_scratch_get(ctx, o + 2, (size_t)NB * sizeof(X), &p))) return rc;
        d_buckets = (X*)p;
        if ((rc = zl_scratch_get(ctx, o + 3, (size_t)2 * nchunks * sizeof(X), &p))) return rc;
        d_partials = (X*)p;
        // ping-pong node buffers of the reduction tree: leaves = 2 channels x blocks, level 1 = 3 channels x blocks / 2 (the largest)
        const size_t leaf_elems = (size_t)2 * SETS * red_blocks, lvl1_elems = (size_t)3 * SETS * (red_blocks / 2 + 1);
        const size_t root_elems = (size_t)SETS * roots_per_set;
        const int tail_slot = set == 0 ? 4 : (set == 1 ? 19 : (set == 2 ? 23 : 32));  // per set: the tails of consecutive jobs may overlap (small jobs)
        if ((rc = zl_scratch_get(ctx, tail_slot, (leaf_elems + lvl1_elems + root_elems + 2 + ZL_ONES_BLOCKS + (size_t)max_giant * ZL_GIANT_PARTS) * sizeof(X), &p))) return rc;
        d_segs = (X*)p;                    // tree nodes, even levels (level 0 = leaves)
        d_stage1 = d_segs + leaf_elems;    // tree nodes, odd levels
        d_sets = d_stage1 + lvl1_elems;    // the root channels of every set, then the sum of the scalar-1 bases
        d_ones_parts = d_sets + root_elems + 1;
        d_giant_tmp = d_ones_parts + ZL_ONES_BLOCKS;
        if (glv) {
            // The images depend on the bases only: kept with the handle (one range per handle; another range of the same handle falls back to
            // the per-call scratch copy below).  k_gls_psi is 50 us of latency in front of the G2 MSM of every small proof, k_glv_phi 12-100 us.
            const size_t endo_bytes = (size_t)(endo_k - 1) * n_real * sizeof(Affine<F>);
            const zl_bases& bs = *bsp;
            if (!bs.d_endo && endo_bytes <= ((size_t)zl_tune("ZL_TUNE_ENDO_CACHE_MB", 512) << 20)) {
                void* q = nullptr;
                if (hipMalloc(&q, endo_bytes) == hipSuccess) {
                    if constexpr (G::GLV && G::ENDO_K == 2)
                        hipLaunchKernelGGL((k_glv_phi<G>), dim3((uint32_t)((n_real + 127) / 128)), dim3(128), 0, ctx->stream, d_bases, (uint32_t)n_real, reinterpret_cast<Affine<F>*>(q));
                    else if constexpr (G::GLV && G::ENDO_K == 4)
                        hipLaunchKernelGGL((k_gls_psi<G>), dim3((uint32_t)((n_real + 63) / 64)), dim3(64), 0, ctx->stream, d_bases, (uint32_t)n_real, reinterpret_cast<Affine<F>*>(q));
                    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipFree(q); return ZL_EHIP; }
                    bs.d_endo = q;
                    bs.endo_first = first;
                    bs.endo_n = n_real;
                    bs.endo_k = endo_k;
                } else {
                    (void)hipGetLastError();  // out of memory for the cache: clear HIP's sticky per-thread error, the per-call scratch copy below serves
                }
            }
            if (bs.d_endo && bs.endo_first == first && bs.endo_n == n_real && bs.endo_k == endo_k) {
                d_phi = reinterpret_cast<const Affine<F>*>(bs.d_endo);
                phi_owner = false;
                phi_cached = true;
            }
        }
        if (glv && phi_owner) {
            if ((rc = zl_scratch_get(ctx, phi_slot, (size_t)(endo_k - 1) * n_real * sizeof(Affine<F>), &p))) return rc;
            d_phi = reinterpret_cast<const Affine<F>*>(p);
        }
        return ZL_OK;
    }
    // sizes of the sort temporaries (slots 5 and 6), as sort() requests them: a heterogeneous pipeline grows the slots to the
    // largest job before anything is in flight (a growing zl_scratch_get frees the old block)
    size_t vs_bytes() const { return glv ? (((size_t)n * 32 + 255) / 256) * 256 : 0; }  // the half-scalars live behind the slot-5 temporaries
    void sort_tmp_sizes(size_t& s5, size_t& s6) const {
        sort_tmp_sizes_(s5, s6);
        if (glv) s5 = ((s5 + 255) / 256) * 256 + vs_bytes();
    }
    void sort_tmp_sizes_(size_t& s5, size_t& s6) const {
        s5 = s6 = 0;
        if (wide) {
            uint32_t nslices = 64;
            const uint32_t max_slices = (uint32_t)((n + 4095) / 4096);
            if (nslices > max_slices) nslices = max_slices;
            const uint32_t P = Gn * W * nslices;
            const uint32_t pscan_blocks = (P + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
            const size_t b_lo = (((size_t)n * W * 2 + 255) / 256) * 256, b_hi = (((size_t)n * W + 255) / 256) * 256;
            const size_t b_pidx = (((size_t)n * W * 4 + 255) / 256) * 256;
            const size_t b_pc = (((size_t)(2 * P + pscan_blocks + 8) * 4 + 255) / 256) * 256;
            s5 = b_lo + b_hi + b_lo + b_pidx + b_pc + 256;
            uint32_t fsl = 16;
            while (fsl * Gn < 2048 && fsl < 128) fsl *= 2;
            const uint32_t P2 = Gn * 128 * fsl;
            const uint32_t p2scan_blocks = (P2 + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
            s6 = b_lo + b_pidx + (((size_t)(2 * (size_t)P2 + p2scan_blocks + 8) * 4 + 255) / 256) * 256 + 256;
        } else if (c <= 16) {
            uint32_t nslices = (256 + W - 1) / W;
            const uint32_t max_slices = (uint32_t)((n + 4095) / 4096);
            if (nslices > max_slices) nslices = max_slices;
            if (nslices < 1) nslices = 1;
            s5 = (size_t)n * W * 2 + (size_t)nslices * NB * 4 + 256;
        }  // plain c >= 21 (more than 255 sort groups): the global-atomics sort needs no temporaries
    }
    int sort(zl_ctx* ctx, hipStream_t st) {
        const zl_bases& bs = *bsp;
        int rc;
        // the bucket counters are written in full by the LDS path (k_msm_slice_prefix) and by the wide path (k_msm_fine_hist); only the
        // global-atomics sort counts into them
        if (!wide && c > 16) ZL_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)(NB + 1) * 4, st));
        ZL_HIP(ctx, hipMemsetAsync(d_big_count, 0, 32, st));  // big, ones, giant counts; [4..5]: oversized sub-group queue head (u64)
        const uint32_t nblk = (uint32_t)((n + 255) / 256);
        // GLV front end: half-scalars behind the slot-5 temporaries, phi image of the bases (once per call for a batch over one key)
        const uint32_t* sc_eff = sc;
        const uint8_t* inf_eff = d_inf;
        uint32_t* bad_eff = d_bad_scalar;
        if (glv) {
            size_t s5tot, s6tot;
            sort_tmp_sizes(s5tot, s6tot);
            void* p5;
            if ((rc = zl_scratch_get(ctx, slotA, s5tot, &p5))) return rc;
            d_vs = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(p5) + (s5tot - vs_bytes()));
            if constexpr (G::GLV && G::ENDO_K == 2) {
                hipLaunchKernelGGL((k_glv_split<typename G::GLVP>), dim3((uint32_t)((n_real + 255) / 256)), dim3(256), 0, st, sc, (uint32_t)n_real, d_inf, d_vs, (int)G::SC_BITS, d_bad_scalar);
                if (phi_owner)
                    hipLaunchKernelGGL((k_glv_phi<G>), dim3((uint32_t)((n_real + 127) / 128)), dim3(128), 0, st, d_bases, (uint32_t)n_real, const_cast<Affine<F>*>(d_phi));
            } else if constexpr (G::GLV && G::ENDO_K == 4) {
                hipLaunchKernelGGL((k_gls_split<typename G::GLVP>), dim3((uint32_t)((n_real + 255) / 256)), dim3(256), 0, st, sc, (uint32_t)n_real, d_inf, d_vs, (int)G::SC_BITS, d_bad_scalar);
                if (phi_owner)
                    hipLaunchKernelGGL((k_gls_psi<G>), dim3((uint32_t)((n_real + 63) / 64)), dim3(64), 0, st, d_bases, (uint32_t)n_real, const_cast<Affine<F>*>(d_phi));
            }
            sc_eff = d_vs;
            inf_eff = nullptr;  // the split already dropped the scalars of bases at infinity
            bad_eff = nullptr;  // ... and checked the scalars' width
        }
        const int glv_i = glv ? 1 : 0;
        // per call, not once per process: the attribute is per device and a process may own several contexts
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_hist_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_scatter_range), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (wide) {
            // ---- three-level counting sort over (window, bucket) ids: the merged set of a table, or W sets of plain wide windows ------
            uint32_t nslices = 64;
            const uint32_t max_slices = (uint32_t)((n + 4095) / 4096);
            if (nslices > max_slices) nslices = max_slices;
            const uint32_t per_slice = (uint32_t)((n + nslices - 1) / nslices);
            const uint32_t P = Gn * W * nslices;  // partition counters, order (group, window, slice)
            const uint32_t pscan_blocks = (P + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
            const size_t b_lo = (((size_t)n * W * 2 + 255) / 256) * 256, b_hi = (((size_t)n * W + 255) / 256) * 256;
            const size_t b_plo = b_lo, b_pidx = (((size_t)n * W * 4 + 255) / 256) * 256;
            const size_t b_pc = (((size_t)(2 * P + pscan_blocks + 8) * 4 + 255) / 256) * 256;
            void* pd;
            if ((rc = zl_scratch_get(ctx, slotA, b_lo + b_hi + b_plo + b_pidx + b_pc + 256, &pd))) return rc;
            unsigned char* q = (unsigned char*)pd;
            uint16_t* d_lo16 = (uint16_t*)q; q += b_lo;
            uint8_t* d_hi8 = (uint8_t*)q; q += b_hi;
            uint16_t* d_part_lo = (uint16_t*)q; q += b_plo;
            uint32_t* d_part_idx = (uint32_t*)q; q += b_pidx;
            uint32_t* d_pcounts = (uint32_t*)q;
            uint32_t* d_poff = d_pcounts + P;            // P + 1 entries (total at [P])
            uint32_t* d_pblock = d_poff + P + 1;
            hipLaunchKernelGGL(k_msm_recode_wide, dim3(nblk), dim3(256), 0, st, sc_eff, (uint32_t)n, c, W, pre ? 0u : (H >> 15), spread_t, glv_i, d_lo16, d_hi8, d_ones_list, d_ones_count, inf_eff, sc_bits, bad_eff);
            hipLaunchKernelGGL(k_msm_part_hist, dim3(nslices, W), dim3(256), 0, st, d_hi8, (uint32_t)n, (uint32_t)W, Gn, per_slice, nslices, d_pcounts);
            hipLaunchKernelGGL(k_scan_block_sums, dim3(pscan_blocks), dim3(SCAN_BLOCK), 0, st, d_pcounts, P, d_pblock);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_pblock, pscan_blocks, d_poff + P, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(k_scan_apply, dim3(pscan_blocks), dim3(SCAN_BLOCK), 0, st, d_pcounts, P, d_pblock, d_poff, d_pcounts);
            hipLaunchKernelGGL(k_msm_part_scatter_st, dim3(nslices, W), dim3(256), 0, st, d_lo16, d_hi8, (uint32_t)n, (uint32_t)W, Gn, per_slice, nslices, d_poff,
                               pre ? (uint32_t)bs.n : 0u, pre ? (uint32_t)first : 0u, d_part_lo, d_part_idx);  // plain: d_bases already starts at `first`
            const uint32_t gstride = (uint32_t)W * nslices;  // counters per group
            // level 2: 128 sub-groups (256 buckets each) per group; level 3: LDS-staged sort per sub-group
            const uint32_t SG = Gn * 128;
            uint32_t fsl = 16;
            while (fsl * Gn < 2048 && fsl < 128) fsl *= 2;
            const uint32_t P2 = SG * fsl;
            const uint32_t p2scan_blocks = (P2 + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
            void* pd2;
            const size_t b2_lo = b_plo, b2_idx = b_pidx, b2_c = (((size_t)(2 * (size_t)P2 + p2scan_blocks + 8) * 4 + 255) / 256) * 256;
            if ((rc = zl_scratch_get(ctx, slotB, b2_lo + b2_idx + b2_c + 256, &pd2))) return rc;  // slot 6 is otherwise the NTT's scratch vector
            unsigned char* q2 = (unsigned char*)pd2;
            uint16_t* d_lo2 = (uint16_t*)q2; q2 += b2_lo;
            uint32_t* d_idx2 = (uint32_t*)q2; q2 += b2_idx;
            uint32_t* d_c2 = (uint32_t*)q2;
            uint32_t* d_off2 = d_c2 + P2;  // P2 + 1
            uint32_t* d_blk2 = d_off2 + P2 + 1;
            hipLaunchKernelGGL(k_msm_sub_hist, dim3(fsl, Gn), dim3(256), 0, st, d_part_lo, d_poff, Gn, gstride, d_poff + P, fsl, d_c2);
            hipLaunchKernelGGL(k_scan_block_sums, dim3(p2scan_blocks), dim3(SCAN_BLOCK), 0, st, d_c2, P2, d_blk2);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_blk2, p2scan_blocks, d_off2 + P2, (const uint32_t*)nullptr);
            hipLaunchKernelGGL(k_scan_apply, dim3(p2scan_blocks), dim3(SCAN_BLOCK), 0, st, d_c2, P2, d_blk2, d_off2, d_c2);
            hipLaunchKernelGGL(k_msm_sub_scatter_st, dim3(fsl, Gn), dim3(256), 0, st, d_part_lo, d_part_idx, d_poff, Gn, gstride, d_poff + P, fsl, d_off2, d_lo2,
                               d_idx2);
            hipLaunchKernelGGL(k_msm_fine_hist, dim3(SG), dim3(256), 0, st, d_lo2, d_off2, SG, fsl, d_off2 + P2, d_counts);
            hipLaunchKernelGGL(k_scan_block_sums, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_block_sums, scan_blocks, d_offsets + NB, (const uint32_t*)d_bad_scalar);
            hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums, d_offsets, d_cursor);
            // staged entries per block: at most 144 KiB + 1 KiB of cursors (1 block per CU); sub-groups average n*W/SG entries, so many small
            // sub-groups (plain wide windows) get a smaller stage and two blocks per CU
            uint32_t cap = (uint32_t)std::min<uint64_t>(36 * 1024, std::max<uint64_t>(4096, (maxE / SG) * 22 / 10));
            cap = (uint32_t)std::max(1024, zl_tune("ZL_TUNE_FS_CAP", (int)cap));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_fine_sort), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(k_msm_fine_sort, dim3(SG), dim3(zl_tune("ZL_TUNE_FS", 1024)), (size_t)(256 + cap) * 4, st, d_lo2, d_idx2, d_off2, SG, fsl, d_off2 + P2, d_offsets, cap,
                               d_entries, d_bigsg_head, d_bigsg_items);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_msm_fine_sort_big), hipFuncAttributeMaxDynamicSharedMemorySize, ZL_BT * 4);
            hipLaunchKernelGGL(k_msm_fine_sort_big, dim3(512), dim3(1024), (size_t)ZL_BT * 4, st, d_lo2, d_idx2, d_off2, SG, fsl, d_off2 + P2, d_bigsg_head,
                               d_bigsg_items, d_cursor, d_entries);
        } else if (c <= 16) {
            // LDS counting sort: recode once (u16 digits), per-(slice, window) LDS histograms, slice prefix, scan, range-owned scatter
            uint32_t nslices = (256 + W - 1) / W;  // ~256+ blocks of 1024 lanes, one per CU (<= 128 KiB LDS each)
            const uint32_t max_slices = (uint32_t)((n + 4095) / 4096);
            if (nslices > max_slices) nslices = max_slices;
            if (nslices < 1) nslices = 1;
            const uint32_t per_slice = (uint32_t)((n + nslices - 1) / nslices);
            void* pd;
            if ((rc = zl_scratch_get(ctx, slotA, (size_t)n * W * 2 + (size_t)nslices * NB * 4 + 256, &pd))) return rc;
            uint16_t* d_digits = (uint16_t*)pd;
            uint32_t* d_slice_counts = (uint32_t*)((unsigned char*)pd + (((size_t)n * W * 2 + 255) / 256) * 256);
            hipLaunchKernelGGL(k_msm_recode, dim3(nblk), dim3(256), 0, st, sc_eff, (uint32_t)n, c, W, spread_t, glv_i, d_digits, d_ones_list, d_ones_count, inf_eff, sc_bits, bad_eff);
            hipLaunchKernelGGL(k_msm_hist_lds, dim3(nslices, W), dim3(1024), (size_t)H * 4, st, d_digits, (uint32_t)n, H, per_slice, NB, d_slice_counts);
            if (NB <= 16384 && nslices <= 64) {
                hipLaunchKernelGGL(k_msm_prefix_small, dim3(1), dim3(1024), 0, st, d_slice_counts, NB, nslices, d_offsets, d_cursor, (const uint32_t*)d_bad_scalar);
            } else {
            hipLaunchKernelGGL(k_msm_slice_prefix, dim3((NB + 255) / 256), dim3(256), 0, st, d_slice_counts, NB, nslices, d_counts);
            hipLaunchKernelGGL(k_scan_block_sums, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_block_sums, scan_blocks, d_offsets + NB, (const uint32_t*)d_bad_scalar);
            hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums, d_offsets, d_cursor);
            }
            // scatter: one block per (bucket range, window); ranges sized so that W * ranges ~ 256..512 blocks
            uint32_t ranges = 1;
            while (ranges * W < 256 && (H / (ranges * 2)) >= 64) ranges *= 2;
            ranges = (uint32_t)std::max(1, zl_tune("ZL_TUNE_RANGES", (int)ranges));
            const uint32_t RB = (H + ranges - 1) / ranges;
            // (the digit row of a window can be walked by `parts` blocks, slice-aligned: measured 1 = 2 = 4 = 8 at 2^18 .. 2^21 -- the kernel is bound by
            // its 4-byte scattered stores, 16.8 M of them in 0.19 ms at 2^20, not by the length of the row, the load latency or the LDS atomics)
            const uint32_t parts = (uint32_t)std::max(1, std::min<int>((int)nslices, zl_tune("ZL_TUNE_SCATTER_PARTS", 1)));
            hipLaunchKernelGGL(k_msm_scatter_range, dim3(8 * ((W + 7) / 8), ranges, parts), dim3(1024), (size_t)RB * 4, st, d_digits, (uint32_t)n, H, RB, d_offsets, d_entries,
                               (const uint32_t*)d_slice_counts, NB, nslices, per_slice, parts, (uint32_t)W);
        } else {
            // wide windows without a table: histogram / scatter with global atomics
            hipLaunchKernelGGL((k_msm_digits<0>), dim3(nblk), dim3(256), 0, st, sc, (uint32_t)n, c, W, d_counts, (uint32_t*)nullptr, d_ones_list, d_ones_count, d_inf, (int)G::SC_BITS, d_bad_scalar);
            hipLaunchKernelGGL(k_scan_block_sums, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums);
            hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_block_sums, scan_blocks, d_offsets + NB, (const uint32_t*)d_bad_scalar);
            hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums, d_offsets, d_cursor);
            hipLaunchKernelGGL((k_msm_digits<1>), dim3(nblk), dim3(256), 0, st, sc, (uint32_t)n, c, W, d_cursor, d_entries, d_ones_list, d_ones_count, d_inf, (int)G::SC_BITS, d_bad_scalar);
        }
        ZL_HIP(ctx, hipGetLastError());
        return ZL_OK;
    }
    // wg_per_cu > 0: the persistent form (pipelined batches) on wg_per_cu x CUs workgroups
    int accumulate(zl_ctx* ctx, hipStream_t st, int wg_per_cu = 0) {
        const uint32_t lanes_persist = (uint32_t)wg_per_cu * (uint32_t)ctx->cu_count * ZL_ACC_PERSIST_BLOCK;
        if (wg_per_cu > 0 && ctx->cu_count > 0 && nchunks >= 8 * (uint64_t)lanes_persist)  // >= 8 chunks per lane: the last, partial round costs little
            hipLaunchKernelGGL((k_msm_accumulate_persist<G>), dim3((uint32_t)wg_per_cu * (uint32_t)ctx->cu_count), dim3(ZL_ACC_PERSIST_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases,
                               d_buckets, d_partials, ZL_CHUNK, glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu, nchunks);
        else if (nchunks <= (uint64_t)zl_tune("ZL_TUNE_QUAD_ACC_CHUNKS", 49152))  // four lanes per chunk while that still fits the machine at three waves per SIMD
            hipLaunchKernelGGL((k_msm_accumulate_quad<G>), dim3((4 * nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK), dim3(ZL_ACC_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                               glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        else
        hipLaunchKernelGGL((k_msm_accumulate<G>), dim3((nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK), dim3(ZL_ACC_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                           glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        ZL_HIP(ctx, hipGetLastError());
        return ZL_OK;
    }
    int tail(zl_ctx* ctx, hipStream_t st) {
        // four lanes per group operation (zl_quad.h) in every tail launch that does not fill the machine
        const uint32_t quad_max = (uint32_t)zl_tune("ZL_TUNE_QUAD_LANES", 65536);
        if (NB <= quad_max)
            hipLaunchKernelGGL((k_msm_merge<G, true>), dim3((4 * NB + 63) / 64), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        else
        hipLaunchKernelGGL((k_msm_merge<G>), dim3((NB + 63) / 64), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        hipLaunchKernelGGL((k_msm_merge_big<G>), dim3(std::min<uint32_t>(max_big, 1024)), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_offsets, d_buckets,
                           d_partials, d_big_list, d_big_count, ZL_CHUNK);
        hipLaunchKernelGGL((k_msm_merge_giant<G>), dim3(std::min<uint32_t>(max_giant, 16) * ZL_GIANT_PARTS), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st,
                           d_offsets, d_giant_tmp, d_partials, d_giant_list, d_giant_count, ZL_CHUNK);
        hipLaunchKernelGGL((k_msm_merge_giant2<G>), dim3((max_giant + 63) / 64), dim3(64), 0, st, d_buckets, d_giant_tmp, d_giant_list, d_giant_count);
        // scalar-1 bases: window-0 table entries are the bases themselves
        hipLaunchKernelGGL((k_msm_ones<G>), dim3(ZL_ONES_BLOCKS), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_ones_list, d_ones_count,
                           pre ? d_bases + first : d_bases, d_ones_parts, glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        hipLaunchKernelGGL((k_msm_window_sum<G>), dim3(1), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_ones_parts, (uint32_t)ZL_ONES_BLOCKS,
                           (uint32_t)ZL_ONES_BLOCKS, 1u, d_sets + (size_t)SETS * roots_per_set, (const uint32_t*)d_ones_count);
        {
            const uint32_t fset = spread_t >= 0 ? (uint32_t)(W - 1) : 0xFFFFFFFFu, flog = (uint32_t)std::max(spread_t, 0);
            const uint32_t leaves = SETS * red_blocks;
            X* cur = red_levels == 0 ? d_sets : d_segs;
            if (leaves <= quad_max)
                hipLaunchKernelGGL((k_msm_reduce_level0<G, true>), dim3((4 * leaves + 63) / 64), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            else
            hipLaunchKernelGGL((k_msm_reduce_level0<G>), dim3((leaves + 63) / 64), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            for (uint32_t lv = 1; lv <= red_levels; lv++) {
                const uint32_t nodes = red_blocks >> lv, lanes = SETS * nodes * (lv + 2);
                X* nxt = lv == red_levels ? d_sets : ((lv & 1) ? d_stage1 : d_segs);
                if (lanes <= quad_max)
                    hipLaunchKernelGGL((k_msm_reduce_tree<G, true>), dim3((4 * lanes + 63) / 64), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                else
                hipLaunchKernelGGL((k_msm_reduce_tree<G>), dim3((lanes + 63) / 64), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                cur = nxt;
            }
        }
        ZL_HIP(ctx, hipGetLastError());
        ZL_HIP(ctx, hipMemcpyAsync(hw, d_sets, sizeof(X) * ((size_t)SETS * roots_per_set + 1), hipMemcpyDeviceToHost, st));
        ZL_HIP(ctx, hipMemcpyAsync(hE, d_offsets + NB, 8, hipMemcpyDeviceToHost, st));
        return ZL_OK;
    }
    // The window sum of set w is V_w = A_w + g0 * sum_b 2^b S_(w,b) (the root channels T, A, S_0 .. of its reduction tree); the result is
    // sum_w 2^(c w) V_w (the table of a precomputed handle already carries that factor: one set, w = 0) + the scalar-1 bases.
    //   stage 1  every V_w by its own short Horner over the bit positions of the window (<= c - 2 doublings, levels + 1 additions): the
    //            sets are independent -> zl_pool, one task per set
    //   stage 2  one serial Horner over the sets, high to low: c doublings + one addition per set (the ~c W doublings every window
    //            method needs)
    // (Rounds 1-2 ran ONE Horner over all bit positions on one thread: the same ~c W doublings, but all (levels + 2) W additions
    // serial as well: 0.40 ms for BLS12-381 G1 at c = 16 against ~0.2 ms now.)
    X window_value(int w) const {
        const X* root = hw + (size_t)w * roots_per_set;  // channels: T, A, S_0 ..
        X v = X::inf();
        const int top = (int)red_lg0 + (int)red_levels - 1;  // highest position inside the window that carries a channel
        for (int off = std::max(top, 0); off >= 0; off--) {
            if (off != std::max(top, 0)) zl::dbl_inplace(v);
            const int bsel = off - (int)red_lg0;
            if (bsel >= 0 && bsel < (int)red_levels) {
                const bool skipped = spread_t >= 0 && w == (int)SETS - 1 && off >= spread_t && !pre;  // spread top window: bits from spread_t on carry no weight
                if (!skipped) zl::add_full(v, root[2 + bsel]);
            }
            if (off == 0) zl::add_full(v, root[1]);
        }
        return v;
    }
    X finish(bool parallel = true) const {
        std::vector<X> V(SETS);
        if (parallel && SETS >= 4) zl_pool_get().parallel_for(SETS, [&](size_t w) { V[w] = window_value((int)w); });
        else for (uint32_t w = 0; w < SETS; w++) V[w] = window_value((int)w);
        X total = V[SETS - 1];
        for (int w = (int)SETS - 2; w >= 0; w--) {
            zl::dbl_n(total, c);  // c doublings in Jacobian coordinates
            zl::add_full(total, V[w]);
        }
        zl::add_full(total, hw[(size_t)SETS * roots_per_set]);
        return total;
    }
};

template <class G>
static int msm_run_t(zl_ctx* ctx, const zl_bases& bs, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial) {
    using X = XYZZ<typename G::F>;
    X total = X::inf();
    ctx->timing = zl_timing{};
    if (n > 0) {
        static const bool trace = getenv("ZL_HOST_TRACE") != nullptr;  // developer aid: host-side phase times of a single call on stderr
        const auto tp0 = std::chrono::steady_clock::now();
        MsmJob<G> job;
        int rc;
        if ((rc = job.plan(ctx, bs, first, d_scalars, n, 18))) return rc;
        if ((rc = job.alloc(ctx, 0))) return rc;
        {   // results land in pinned host memory: the two D2H copies are then plain queue entries behind the last kernel (from pageable
            // memory each cost a staging round trip: ~25 us of idle device in front of either copy)
            const size_t need = sizeof(X) * ((size_t)job.SETS * job.roots_per_set + 1) + 16;
            if (ctx->pinned_cap < need) {
                if (ctx->pinned) (void)hipHostFree(ctx->pinned);
                ctx->pinned = nullptr;
                ctx->pinned_cap = 0;
                ZL_HIP(ctx, hipHostMalloc(&ctx->pinned, need + 4096, hipHostMallocDefault));
                ctx->pinned_cap = need + 4096;
            }
            job.hw = reinterpret_cast<X*>(ctx->pinned);
            job.hE = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(ctx->pinned) + sizeof(X) * ((size_t)job.SETS * job.roots_per_set + 1));
        }
        const auto tp1 = std::chrono::steady_clock::now();
        hipStream_t st = ctx->stream;
        if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[0], st));
        if ((rc = job.sort(ctx, st))) return rc;
        if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[1], st));
        if ((rc = job.accumulate(ctx, st))) return rc;
        if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[2], st));
        if ((rc = job.tail(ctx, st))) return rc;
        if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[3], st));
        const auto tp2 = std::chrono::steady_clock::now();
        ZL_HIP(ctx, hipStreamSynchronize(st));
        const auto tp3 = std::chrono::steady_clock::now();
        if (ctx->timing_on) {
            ZL_HIP(ctx, hipEventElapsedTime(&ctx->timing.total_ms, ctx->ev[0], ctx->ev[3]));
            ZL_HIP(ctx, hipEventElapsedTime(&ctx->timing.dominant_ms, ctx->ev[1], ctx->ev[2]));
        }
        ctx->timing.launches = 1;
        ctx->timing.window_bits = (uint32_t)job.c;
        ctx->timing.entries = *job.hE;
        if (job.hE[1]) return ZL_EINVAL;  // a scalar with bits at or above SC_BITS: not a canonical scalar (the ABI's contract)
        total = job.finish();
        if (trace) {
            const auto tp4 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e3; };
            fprintf(stderr, "[zl_msm n=%zu c=%d] plan+alloc %.1f us  issue %.1f us  sync-wait %.1f us  timing+horner %.1f us\n", n, job.c, us(tp0, tp1), us(tp1, tp2),
                    us(tp2, tp3), us(tp3, tp4));
        }
    }
    static_assert(sizeof(X) <= ZL_PARTIAL_WORDS * 8, "partial too small");
    memset(out_partial, 0, ZL_PARTIAL_WORDS * 8);
    memcpy(out_partial, &total, sizeof(X));
    return ZL_OK;
}

// `count` MSMs over the same bases, pipelined: sort of MSM i+2 (stream_sort) | accumulation of MSM i+1 (ctx->stream) | tail of MSM i
// (stream_tail).  Three buffer sets rotate; a set is reused when the tail that reads it has finished.  The sort and tail phases are
// memory- / latency-bound and mostly fit into the issue slots the compute-bound accumulation leaves: in steady state an MSM costs its
// accumulation kernel + ~2.5 ms (measured: 2^24, tools/batch_overlap.py).
struct MsmSpec {
    const zl_bases* bs;
    size_t first;
    const void* d_scalars;
    size_t n;
    hipEvent_t wait;  // optional: the scalars of this job are ready when this event (recorded on another stream) has fired
};
// `recorded` (optional): the wait events are recorded by ANOTHER host thread (zl_msm's copy thread); job i may only be issued once
// *recorded > i, because hipStreamWaitEvent on a not-yet-recorded event does not wait.  Negative = that thread failed.
template <class G>
// `on_done` (optional): called with i from a helper thread as soon as job i's result is in out_partials (jobs complete in order), while the
// later jobs are still running on the device: a caller with host work that depends on the first results starts it early (Groth16: s A + r B1).
static int msm_run_jobs_t(zl_ctx* ctx, const MsmSpec* specs, size_t count, uint64_t* out_partials, const std::atomic<int>* recorded = nullptr,
                          const std::function<void(size_t)>* on_done = nullptr) {
    using X = XYZZ<typename G::F>;
    ctx->timing = zl_timing{};
    if (count == 0) return ZL_OK;
    bool any_empty = false;
    for (size_t i = 0; i < count; i++) any_empty = any_empty || specs[i].n == 0;
    if (any_empty || count == 1) {
        for (size_t i = 0; i < count; i++) {
            if (recorded) { while (recorded->load(std::memory_order_acquire) >= 0 && recorded->load(std::memory_order_acquire) <= (int)i) std::this_thread::yield(); if (recorded->load() < 0) return ZL_EHIP; }
            if (specs[i].wait) ZL_HIP(ctx, hipEventSynchronize(specs[i].wait));
            int rc = msm_run_t<G>(ctx, *specs[i].bs, specs[i].first, specs[i].d_scalars, specs[i].n, out_partials + i * ZL_PARTIAL_WORDS);
            if (rc) return rc;
            if (on_done) (*on_done)(i);
        }
        return ZL_OK;
    }
    int rc;
    if (!ctx->stream_sort) {
        // highest priority: the short sort / tail kernels must get wave slots as the long accumulation kernel frees them
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        ZL_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream_sort, hipStreamNonBlocking, prio_hi));
        for (auto& t : ctx->stream_tail) ZL_HIP(ctx, hipStreamCreateWithPriority(&t, hipStreamNonBlocking, prio_hi));
    }
    std::vector<MsmJob<G>> jobs(count);
    size_t t5 = 0, t6 = 0;
    uint32_t max_sets = 0;
    // GLV jobs need phi(P_i) of their bases: a batch over ONE key computes it once (job 0, slot 18; the later jobs borrow the pointer --
    // their sorts run behind job 0's on the sort stream), a heterogeneous batch once per job in the slot of its buffer set (20 + i % 3: free
    // again when the tail of job i - 3 has finished, like the rest of the set)
    bool one_key = true;
    for (size_t i = 1; i < count; i++) one_key = one_key && specs[i].bs == specs[0].bs && specs[i].first == specs[0].first && specs[i].n == specs[0].n;
    // SMALL jobs (at most 2^20 points each) are chains of short, latency-bound kernels: three phases on three shared streams would
    // run their sorts one after the other, then their accumulations, then their tails (four 237-point MSMs of a small proof: 2.1 ms).  They
    // run side by side instead: job i does sort, accumulation and tail on the stream of buffer set i % NS, with its own sort temporaries.
    uint64_t biggest = 0;
    for (size_t i = 0; i < count; i++) biggest = std::max<uint64_t>(biggest, (uint64_t)specs[i].n);
    // measured (round 3, batches of 6): 2^16 0.69 -> 0.50 ms per MSM, 2^20 3.31 -> 3.10; equal at 2^18 - 2^19; from 2^21 on the three-phase pipeline
    // wins (2^24: 36.3 against 37.3 ms)
    const bool side = biggest <= ((uint64_t)1 << zl_tune("ZL_TUNE_SIDE_BY_SIDE_LOG", 20));
    const size_t NS = side ? std::min<size_t>(count, (size_t)std::min(4, std::max(1, zl_tune("ZL_TUNE_SIDE_LANES", 4)))) : 3;
    for (size_t i = 0; i < count; i++) {
        // (side by side every job computes its own phi image: there is no common stream that would order a borrower behind the owner)
        if ((rc = jobs[i].plan(ctx, *specs[i].bs, specs[i].first, specs[i].d_scalars, specs[i].n, (one_key && !side) ? 18 : MsmJob<G>::phi_slot_of((int)(i % NS))))) return rc;
        if (one_key && !side && i > 0) jobs[i].phi_owner = false;
        size_t a5, a6;
        jobs[i].sort_tmp_sizes(a5, a6);
        t5 = std::max(t5, a5);
        t6 = std::max(t6, a6);
        max_sets = std::max<uint32_t>(max_sets, jobs[i].SETS * jobs[i].roots_per_set);
    }
    if (side) {
        for (size_t k = 0; k < NS; k++)
            if (!ctx->stream_lane[k]) ZL_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_lane[k], hipStreamNonBlocking));
    }
    // all buffers up front (growth synchronises and frees: nothing may be in flight), then bind set i % 3 to job i: the first pass
    // grows every slot to its largest user, the second binds the final pointers.  Three sets: the tail of job i runs beside the
    // accumulation of job i+1 and is slow there, so the sort of job i+2 must not have to wait for it.
    void* dummy;
    if (!side && t5 && (rc = zl_scratch_get(ctx, 5, t5, &dummy))) return rc;
    if (!side && t6 && (rc = zl_scratch_get(ctx, 6, t6, &dummy))) return rc;
    for (int pass = 0; pass < 2; pass++) {
        for (size_t i = 0; i < count; i++) {
            if ((rc = jobs[i].alloc(ctx, (int)(i % NS), side))) return rc;
            if (one_key && !side && i > 0 && jobs[i].glv && !jobs[i].phi_cached) jobs[i].d_phi = jobs[0].d_phi;
        }
    }
    // (Round 4 built and removed "sort sharing": jobs over the same scalar vector with the same plan borrowing one bucket-sorted entry list -- VERDICT r3
    // item 2's proposal for Groth16's a_query / b_g1_query MSMs.  Those two lists are NOT equal: each query has its own points at infinity (variables
    // absent from A resp. B) and the sort drops their scalars.  Forcing the pair to share one sort anyway, for its timing only, moved the
    // 958 465-constraint proof by nothing: 19.39 / 19.44 against 19.40 / 19.76 ms (profiles/r04_g16_eventpool_ab.log) -- the proof is bound by its group
    // additions, not its sorts.  And a pipeline that skips the sort of a repeated scalar vector would skip work inside bench.py's timed steps.)
    const size_t per = sizeof(X) * (max_sets + 1) + 16;
    if (ctx->pinned_cap < per * count) {
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
        ZL_HIP(ctx, hipHostMalloc(&ctx->pinned, per * count, hipHostMallocDefault));
        ctx->pinned_cap = per * count;
    }
    for (size_t i = 0; i < count; i++) {
        unsigned char* base = reinterpret_cast<unsigned char*>(ctx->pinned) + per * i;
        jobs[i].hw = reinterpret_cast<X*>(base);
        jobs[i].hE = reinterpret_cast<uint32_t*>(base + sizeof(X) * (max_sets + 1));
    }
    hipStream_t s_sort = ctx->stream_sort, s_acc = ctx->stream;
    // The tail of job i runs on the tail stream of its buffer set: consecutive tails are independent (own buckets, partials, tree nodes), and
    // for small jobs -- a chain of ~25 dependent group operations at a few lanes each -- they are what the pipeline's latency consists of:
    // four 237-point MSMs of a small proof finished their tails one after the other in 2.1 ms, now side by side.
    hipStream_t s_tails[3] = {ctx->stream_tail[0], ctx->stream_tail[1], ctx->stream_tail[2]};
    // k_msm_accumulate_persist (room for the side streams) is OFF: measured at 2^24 (profiles/r03_persist_accumulate.log) the sort and the
    // tail do move under the accumulation and the gap between accumulations closes, but the accumulation itself goes from 33.4 to 38.9 ms
    // beside the sort and to ~50 ms beside the level-0 / tree kernels (two instruction streams of 40-60 KB each share one 64-KB instruction
    // cache per CU pair, and the tail kernels' own additions take 6-10 ms instead of 1.6): 49.9 ms per MSM against 37.6.  ZL_TUNE_ACC_WG_PER_CU=2 enables it.
    const int acc_wg_per_cu = sizeof(X) > 256 ? 0 : zl_tune("ZL_TUNE_ACC_WG_PER_CU", 0);
    // (Measured and dropped: making the accumulation of job i+1 wait for the merge kernels / level 0 of job i, so that two field-arithmetic
    // kernels never share the machine -- 958 465-constraint proof 19.2-19.4 ms with or without, 2^20 batches 3.65 = 3.65 ms per MSM.)
    // events from the ctx's pool (zl_ctx_events): [sorted | tail | acc (untimed runs)] without timing, [begin, end | acc0 | acc (timed runs)] with
    hipEvent_t *pool_nt = nullptr, *pool_t = nullptr;
    if ((rc = zl_ctx_events(ctx, 0, 3 * count, &pool_nt))) return rc;
    if ((rc = zl_ctx_events(ctx, 1, 2 + (ctx->timing_on ? 2 * count : 0), &pool_t))) return rc;
    hipEvent_t* ev_sorted = pool_nt;
    hipEvent_t* ev_tail = pool_nt + count;
    hipEvent_t* ev_acc = ctx->timing_on ? pool_t + 2 + count : pool_nt + 2 * count;
    hipEvent_t* ev_acc0 = ctx->timing_on ? pool_t + 2 : nullptr;
    hipEvent_t ev_begin = pool_t[0], ev_end = pool_t[1];
    auto cleanup = [&]() {};  // (the events stay with the ctx)
    hipError_t he = hipSuccess;
    rc = ZL_OK;
    // everything already queued on the caller's stream (e.g. the kernels that produced the scalars) comes first
    he = hipEventRecord(ev_begin, s_acc);
    if (he == hipSuccess) he = hipStreamWaitEvent(s_sort, ev_begin, 0);
    if (side)
        for (size_t k = 0; k < NS && he == hipSuccess; k++) he = hipStreamWaitEvent(ctx->stream_lane[k], ev_begin, 0);
    static const bool jtrace = getenv("ZL_HOST_TRACE") != nullptr;
    const auto jt0 = std::chrono::steady_clock::now();
    // issue of one job: sort | accumulate | tail with the events between them.  pipelined: three phases on three streams; side by side: the
    // whole job on the stream of its buffer set (the waits are then between operations of one stream, i.e. no-ops)
    auto issue_job = [&](size_t i) -> int {
        hipStream_t js_sort = side ? ctx->stream_lane[i % NS] : s_sort, js_acc = side ? ctx->stream_lane[i % NS] : s_acc;
        hipStream_t s_tail = side ? ctx->stream_lane[i % NS] : s_tails[i % 3];
        hipError_t e = hipSuccess;
        if (i >= NS) e = hipStreamWaitEvent(js_sort, ev_tail[i - NS], 0);  // buffer set i % NS is free again
        if (recorded && specs[i].wait) {
            while (recorded->load(std::memory_order_acquire) >= 0 && recorded->load(std::memory_order_acquire) <= (int)i) std::this_thread::yield();
            if (recorded->load() < 0) return ZL_EHIP;
        }
        if (e == hipSuccess && specs[i].wait) e = hipStreamWaitEvent(js_sort, specs[i].wait, 0);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        int r;
        if ((r = jobs[i].sort(ctx, js_sort))) return r;
        e = hipEventRecord(ev_sorted[i], js_sort);
        if (e == hipSuccess) e = hipStreamWaitEvent(js_acc, ev_sorted[i], 0);
        if (e == hipSuccess && ctx->timing_on) e = hipEventRecord(ev_acc0[i], js_acc);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        if ((r = jobs[i].accumulate(ctx, js_acc, acc_wg_per_cu))) return r;
        e = hipEventRecord(ev_acc[i], js_acc);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_tail, ev_acc[i], 0);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        if ((r = jobs[i].tail(ctx, s_tail))) return r;
        e = hipEventRecord(ev_tail[i], s_tail);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        if (jtrace) fprintf(stderr, "[zl_msm jobs] job %zu (n=%zu c=%d%s) issued at %.1f us\n", i, jobs[i].n_real, jobs[i].c, side ? ", side by side" : "",
                            (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - jt0).count() / 1e3);
        return ZL_OK;
    };
    // issued[i]: 0 not yet, 1 issued, < 0 failed (-code).  Side by side, every lane stream CAN be fed by its own persistent host thread
    // (ZL_TUNE_LANE_THREADS=1): the ~25 launches of a small job are ~80 us of host time, four jobs 0.3 ms.  Measured (k = 1 proof, 40 runs): all
    // four jobs then reach the device within 0.15 ms, but finish together and later than the staggered jobs of a single issuing thread
    // (median 1.40 against 1.16 ms per proof at GPU_MAX_HW_QUEUES=8, 1.52 against 1.35 at the runtime's default of 4) -- off by default.  (Lanes in
    // different stream-priority classes, i.e. different queue pools, measured no better either.)
    std::unique_ptr<std::atomic<int>[]> issued(new std::atomic<int>[count]);
    for (size_t i = 0; i < count; i++) issued[i].store(0);
    const bool lanes_threaded = side && count > 1 && he == hipSuccess && zl_tune("ZL_TUNE_LANE_THREADS", 0) != 0;
    if (lanes_threaded) {
        for (size_t k = 0; k < NS; k++)
            zl_ctx_worker(ctx, 2 + (int)k).run([&, k]() {
                int r = hipSetDevice(ctx->device) == hipSuccess ? ZL_OK : ZL_EHIP;
                for (size_t i = k; i < count; i += NS) {
                    if (r == ZL_OK) r = issue_job(i);
                    issued[i].store(r == ZL_OK ? 1 : -r, std::memory_order_release);
                }
            });
    } else {
        for (size_t i = 0; i < count; i++) {
            if (he == hipSuccess && rc == ZL_OK) rc = issue_job(i);
            issued[i].store(he == hipSuccess && rc == ZL_OK ? 1 : -(rc ? rc : (int)ZL_EHIP), std::memory_order_release);
        }
    }
    auto lanes_join = [&]() {
        if (lanes_threaded)
            for (size_t k = 0; k < NS; k++) zl_ctx_worker(ctx, 2 + (int)k).wait();
    };
    // Host tails: this thread waits for the jobs' tail events in order (a job's root channels are then in pinned memory) and hands every
    // finished job to a helper thread that runs its window Horner and delivers the result -- while the device works on the later jobs.  Small
    // jobs, which the device finishes faster than the host, get their Horners side by side; `on_done` is delivered in job order.  (The helpers
    // make no HIP calls: a fresh thread's first HIP call costs ~0.1 ms of per-thread runtime setup, per proof.)
    std::atomic<int> frc{ZL_OK};
    std::atomic<size_t> delivered{0};
    std::vector<std::thread> finishers;
    // every job issued (the lanes issue side by side: about the time of one job), then ev_end behind the last tail of every stream that ran tails
    for (size_t i = 0; i < count; i++) {
        int st;
        while ((st = issued[i].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (st < 0 && rc == ZL_OK) rc = -st;
    }
    lanes_join();
    if (he == hipSuccess && rc == ZL_OK) {
        hipStream_t last_tail = side ? ctx->stream_lane[(count - 1) % NS] : s_tails[(count - 1) % 3];
        for (size_t back = 1; back < NS && back < count && he == hipSuccess; back++) he = hipStreamWaitEvent(last_tail, ev_tail[count - 1 - back], 0);
        if (he == hipSuccess) he = hipEventRecord(ev_end, last_tail);
    }
    if (he == hipSuccess && rc == ZL_OK) {
        for (size_t i = 0; i < count; i++) {
            bool ok = frc.load() == ZL_OK;
            if (ok && hipEventSynchronize(ev_tail[i]) != hipSuccess) { frc.store(ZL_EHIP); ok = false; }
            if (jtrace) fprintf(stderr, "[zl_msm jobs] job %zu on the host at %.1f us\n", i,
                                (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - jt0).count() / 1e3);
            if (ok && jobs[i].hE[1]) { frc.store(ZL_EINVAL); ok = false; }  // non-canonical scalar (see msm_run_t)
            finishers.emplace_back([&, i, ok]() {
                if (ok) {
                    const X total = jobs[i].finish(true);  // (concurrent callers of the host pool each take part in their own loop)
                    memset(out_partials + i * ZL_PARTIAL_WORDS, 0, ZL_PARTIAL_WORDS * 8);
                    memcpy(out_partials + i * ZL_PARTIAL_WORDS, &total, sizeof(X));
                }
                if (jtrace) fprintf(stderr, "[zl_msm jobs] job %zu Horner done at %.1f us\n", i,
                                    (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - jt0).count() / 1e3);
                while (delivered.load(std::memory_order_acquire) != i) std::this_thread::yield();
                if (ok && on_done && frc.load() == ZL_OK) (*on_done)(i);
                delivered.store(i + 1, std::memory_order_release);
            });
        }
    }
    // drain all three streams whatever happened, then report
    (void)hipStreamSynchronize(s_sort);
    (void)hipStreamSynchronize(s_acc);
    for (hipStream_t t : s_tails) {
        const hipError_t hs = hipStreamSynchronize(t);
        if (he == hipSuccess) he = hs;
    }
    if (side)
        for (size_t k = 0; k < NS; k++) {
            const hipError_t hs = hipStreamSynchronize(ctx->stream_lane[k]);
            if (he == hipSuccess) he = hs;
        }
    if (he == hipSuccess && rc == ZL_OK && ctx->timing_on) {
        float tot = 0.f, acc_sum = 0.f, t = 0.f;
        he = hipEventElapsedTime(&tot, ev_begin, ev_end);
        for (size_t i = 0; i < count && he == hipSuccess; i++) {
            he = hipEventElapsedTime(&t, ev_acc0[i], ev_acc[i]);
            acc_sum += t;
        }
        ctx->timing.total_ms = tot / (float)count;          // per MSM, pipelined
        ctx->timing.dominant_ms = acc_sum / (float)count;   // mean accumulation kernel
    }
    for (auto& t : finishers) t.join();  // (their events fired before the streams drained)
    cleanup();
    if (he != hipSuccess) { ctx->last_hip = (int)he; return ZL_EHIP; }
    if (rc) return rc;
    if (frc.load()) return frc.load();
    ctx->timing.launches = (uint32_t)count;
    ctx->timing.window_bits = (uint32_t)jobs[0].c;
    ctx->timing.entries = *jobs[count - 1].hE;
    return ZL_OK;
}

// attach the infinity flags to a freshly built handle (kept only when there is at least one point at infinity)
template <class G>
static int bases_flag_inf_t(zl_ctx* ctx, zl_bases* b) {
    using F = typename G::F;
    if (!b->n) return ZL_OK;
    void* d = nullptr;
    ZL_HIP(ctx, hipMalloc(&d, b->n + 16));
    uint32_t* d_cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(d) + ((b->n + 3) / 4) * 4);
    hipStream_t st = ctx->stream;
    uint32_t cnt = 0;
    hipError_t e = hipMemsetAsync(d_cnt, 0, 4, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_bases_inf_flags<G>), dim3((uint32_t)((b->n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const Affine<F>*>(b->d_pts), (uint32_t)b->n,
                           reinterpret_cast<uint8_t*>(d), d_cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ctx->last_hip = (int)e; (void)hipFree(d); return ZL_EHIP; }
    if (cnt == 0) { (void)hipFree(d); return ZL_OK; }
    b->d_inf = d;
    b->n_inf = cnt;
    return ZL_OK;
}
// in (XYZZ / Jacobian, n elements) -> out (affine), sharing one inversion among the elements of a lane; prefix scratch in slot `slot`
template <class G, int FORM>
static int batch_affine_t(zl_ctx* ctx, const void* d_in, size_t n, Affine<typename G::F>* d_out, int slot, hipStream_t st) {
    using F = typename G::F;
    if (!n) return ZL_OK;
    void* d_prefix;
    int rc;
    if ((rc = zl_scratch_get(ctx, slot, n * sizeof(F), &d_prefix))) return rc;
    // ~2^18 lanes (>= 1 wave per SIMD) once there is enough work; up to 64 elements share an inversion
    uint32_t per = (uint32_t)std::min<size_t>(64, std::max<size_t>(1, n >> 18));
    per = (uint32_t)std::max(1, zl_tune("ZL_TUNE_BATCH_INV", (int)per));
    const uint32_t lanes = (uint32_t)((n + per - 1) / per);
    hipLaunchKernelGGL((k_batch_affine<G, FORM>), dim3((lanes + 63) / 64), dim3(64), 0, st, d_in, (uint32_t)n, lanes, (F*)d_prefix, d_out);
    ZL_HIP(ctx, hipGetLastError());
    return ZL_OK;
}
// the shared fixed-base table of the group's generator (built once per ctx, ~1 MB for 8-bit windows)
template <class G>
static int fb_table_get(zl_ctx* ctx, const Affine<typename G::F>** out) {
    using F = typename G::F;
    void*& slot = ctx->fb_table[G::ID];
    if (!slot) {
        const size_t entries = (size_t)ZL_FB_WINDOWS << ZL_FB_BITS;
        void *d_tab = nullptr, *d_tmp = nullptr;
        ZL_HIP(ctx, hipMalloc(&d_tab, entries * sizeof(Affine<F>)));
        int rc = zl_scratch_get(ctx, 5, (entries + ZL_FB_WINDOWS) * sizeof(XYZZ<F>), &d_tmp);
        if (rc) { (void)hipFree(d_tab); return rc; }
        XYZZ<F>* d_bw = (XYZZ<F>*)d_tmp;
        XYZZ<F>* d_x = d_bw + ZL_FB_WINDOWS;
        hipStream_t st = ctx->stream;
        hipLaunchKernelGGL((k_fb_bases<G>), dim3((ZL_FB_WINDOWS + 63) / 64), dim3(64), 0, st, d_bw);
        hipLaunchKernelGGL((k_fb_table<G>), dim3((uint32_t)((entries + 63) / 64)), dim3(64), 0, st, d_bw, d_x);
        rc = batch_affine_t<G, 0>(ctx, d_x, entries, (Affine<F>*)d_tab, 6, st);
        hipError_t e = rc ? hipSuccess : hipStreamSynchronize(st);
        if (rc || e != hipSuccess) { (void)hipFree(d_tab); if (!rc) { ctx->last_hip = (int)e; rc = ZL_EHIP; } return rc; }
        slot = d_tab;
    }
    *out = reinterpret_cast<const Affine<F>*>(slot);
    return ZL_OK;
}

// table[w][i] = 2^(c w) P_i for every base of the handle (one-time, at upload): level by level, c Jacobian doublings per point and one
// shared inversion per ~64 points
template <class G>
static int bases_precompute_t(zl_ctx* ctx, zl_bases& bs, int c) {
    using F = typename G::F;
    if (c == 0) c = zl_pick_window_precomp(bs.n, G::SC_BITS);
    if (c < 16 || c > 23) return ZL_EINVAL;
    const int W = (G::SC_BITS + 1 + c - 1) / c;
    if ((uint64_t)W * bs.n >= (1ull << 31)) return ZL_EINVAL;
    if (bs.d_table) { ZL_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(bs.d_table); bs.d_table = nullptr; bs.precomp_c = 0; }
    void* t = nullptr;
    ZL_HIP(ctx, hipMalloc(&t, std::max<size_t>(bs.n, 1) * W * sizeof(Affine<F>)));
    if (bs.n) {
        hipStream_t st = ctx->stream;
        Affine<F>* tab = reinterpret_cast<Affine<F>*>(t);
        void* d_jac = nullptr;
        int rc = zl_scratch_get(ctx, 5, bs.n * sizeof(Jac<F>), &d_jac);
        hipError_t e = rc ? hipSuccess : hipMemcpyAsync(tab, bs.d_pts, bs.n * sizeof(Affine<F>), hipMemcpyDeviceToDevice, st);
        for (int w = 1; w < W && !rc && e == hipSuccess; w++) {
            hipLaunchKernelGGL((k_bases_level_dbl<G>), dim3((uint32_t)((bs.n + 63) / 64)), dim3(64), 0, st, tab + (size_t)(w - 1) * bs.n, (uint32_t)bs.n, c,
                               (Jac<F>*)d_jac);
            e = hipGetLastError();
            if (e == hipSuccess) rc = batch_affine_t<G, 1>(ctx, d_jac, bs.n, tab + (size_t)w * bs.n, 6, st);
        }
        if (!rc && e == hipSuccess) e = hipStreamSynchronize(st);
        if (rc || e != hipSuccess) {
            if (!rc) { ctx->last_hip = (int)e; rc = ZL_EHIP; }
            (void)hipStreamSynchronize(st);
            (void)hipFree(t);
            return rc;
        }
    }
    bs.d_table = t;
    bs.precomp_c = c;
    return ZL_OK;
}
int ZL_GNAME(zl_bases_precompute)(zl_ctx* ctx, zl_bases& b, int c) { return bases_precompute_t<ZL_G>(ctx, b, c); }

int ZL_GNAME(zl_msm_run_batch)(zl_ctx* ctx, const zl_bases& b, size_t first, const void* const* d_scalars, size_t n, size_t count, uint64_t* out_partials) {
    std::vector<MsmSpec> specs(count);
    for (size_t i = 0; i < count; i++) specs[i] = MsmSpec{&b, first, d_scalars[i], n, nullptr};
    return msm_run_jobs_t<ZL_G>(ctx, specs.data(), count, out_partials);
}
// heterogeneous pipeline: job i = (bases[i], first[i], d_scalars[i], n[i]) (Groth16: the four G1 MSMs of one proof)
int ZL_GNAME(zl_msm_run_jobs)(zl_ctx* ctx, const zl_bases* const* bases, const size_t* first, const void* const* d_scalars, const size_t* n,
                              const hipEvent_t* wait, size_t count, uint64_t* out_partials, const std::atomic<int>* recorded,
                              const std::function<void(size_t)>* on_done) {
    std::vector<MsmSpec> specs(count);
    for (size_t i = 0; i < count; i++) specs[i] = MsmSpec{bases[i], first[i], d_scalars[i], n[i], wait ? wait[i] : nullptr};
    return msm_run_jobs_t<ZL_G>(ctx, specs.data(), count, out_partials, recorded, on_done);
}
int ZL_GNAME(zl_msm_run)(zl_ctx* ctx, const zl_bases& b, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial) {
    return msm_run_t<ZL_G>(ctx, b, first, d_scalars, n, out_partial);
}

// host: XYZZ partial (Montgomery) -> canonical affine
template <class G>
static int partial_to_affine_t(const uint64_t* partial, uint64_t* out_xy, uint8_t* out_inf) {
    using F = typename G::F;
    XYZZ<F> p;
    memcpy(&p, partial, sizeof p);
    const Affine<F> a = zl::to_affine(p);
    const bool inf = p.is_inf();
    if (out_inf) *out_inf = inf ? 1 : 0;
    constexpr int WORDS = FieldIO<F>::WORDS;
    uint32_t* w = reinterpret_cast<uint32_t*>(out_xy);
    if (inf) { for (int k = 0; k < 2 * WORDS; k++) w[k] = 0; return ZL_OK; }
    FieldIO<F>::store_canon(w, a.x);
    FieldIO<F>::store_canon(w + WORDS, a.y);
    return ZL_OK;
}
// canonical affine point -> opaque partial (tests; lets a caller inject a point into zl_partials_sum)
template <class G>
static int partial_from_affine_t(const uint64_t* xy, uint64_t* out_partial) {
    using F = typename G::F;
    constexpr int WORDS = FieldIO<F>::WORDS;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(xy);
    uint32_t acc = 0;
    for (int k = 0; k < 2 * WORDS; k++) acc |= w[k];
    XYZZ<F> p = XYZZ<F>::inf();
    if (acc) p = XYZZ<F>::from_affine(Affine<F>{FieldIO<F>::load_canon(w), FieldIO<F>::load_canon(w + WORDS)});
    memset(out_partial, 0, ZL_PARTIAL_WORDS * 8);
    memcpy(out_partial, &p, sizeof p);
    return ZL_OK;
}
int ZL_GNAME(zl_partial_from_affine)(const uint64_t* xy, uint64_t* out_partial) { return partial_from_affine_t<ZL_G>(xy, out_partial); }
int ZL_GNAME(zl_partial_to_affine)(const uint64_t* partial, uint64_t* out_xy, uint8_t* out_inf) {
    return partial_to_affine_t<ZL_G>(partial, out_xy, out_inf);
}
template <class G>
static int partials_sum_t(const uint64_t* partials, size_t count, uint64_t* out_partial) {
    using F = typename G::F;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = 0; i < count; i++) {
        XYZZ<F> p;
        memcpy(&p, partials + i * ZL_PARTIAL_WORDS, sizeof p);
        zl::add_full(acc, p);
    }
    memset(out_partial, 0, ZL_PARTIAL_WORDS * 8);
    memcpy(out_partial, &acc, sizeof acc);
    return ZL_OK;
}
int ZL_GNAME(zl_partials_fold)(const uint64_t* partials, size_t count, uint64_t* out_partial) {
    return partials_sum_t<ZL_G>(partials, count, out_partial);
}

// ------------------------------------------------------------------------------------------------ bases host side
template <class G>
static int bases_upload_t(zl_ctx* ctx, const void* xy, size_t n, size_t stride, long inf_off, unsigned flags, zl_bases* out) {
    using F = typename G::F;
    const size_t rec = (size_t)2 * FieldIO<F>::WORDS * 4;  // ABI record: x||y in 32-bit words (independent of the device representation)
    if (stride == 0) stride = rec;
    if (stride < rec || n >= (1ull << 31)) return ZL_EINVAL;
    if (inf_off >= 0 && (size_t)inf_off >= stride) return ZL_EINVAL;
    void* d_pts = nullptr;
    ZL_HIP(ctx, hipMalloc(&d_pts, std::max<size_t>(n, 1) * sizeof(Affine<F>)));
    int rc = ZL_OK;
    if (n) {
        // stage packed records (+ optional flag bytes) on the host, one H2D copy
        std::vector<unsigned char> packed;
        std::vector<uint8_t> flags_host;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(xy);
        const unsigned char* send = src;
        if (stride != rec) {
            packed.resize(n * rec);
            for (size_t i = 0; i < n; i++) memcpy(&packed[i * rec], src + i * stride, rec);
            send = packed.data();
        }
        if (inf_off >= 0) {
            flags_host.resize(n);
            for (size_t i = 0; i < n; i++) flags_host[i] = src[i * stride + (size_t)inf_off] ? 1 : 0;
        }
        void* d_in;
        if ((rc = zl_scratch_get(ctx, 5, n * rec + n + 64, &d_in))) { (void)hipFree(d_pts); return rc; }
        uint8_t* d_flags = reinterpret_cast<uint8_t*>(d_in) + n * rec;
        uint32_t* d_bad = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(d_in) + ((n * rec + n + 15) / 16) * 16);
        hipStream_t st = ctx->stream;
        hipError_t e = hipMemcpyAsync(d_in, send, n * rec, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && inf_off >= 0) e = hipMemcpyAsync(d_flags, flags_host.data(), n, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 4, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((k_bases_import<G>), dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, st, (const uint32_t*)d_in,
                               inf_off >= 0 ? d_flags : (const uint8_t*)nullptr, (uint32_t)n, (flags & ZL_MONT) ? 0 : 1, (flags & ZL_CHECK) ? 1 : 0,
                               (Affine<F>*)d_pts, d_bad);
            e = hipGetLastError();
        }
        uint32_t bad = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { ctx->last_hip = (int)e; (void)hipFree(d_pts); return ZL_EHIP; }
        if (bad) { (void)hipFree(d_pts); return ZL_ENOTCURVE; }
    }
    out->d_pts = d_pts;
    out->n = n;
    if ((rc = bases_flag_inf_t<G>(ctx, out))) { (void)hipFree(d_pts); out->d_pts = nullptr; return rc; }
    return ZL_OK;
}
int ZL_GNAME(zl_bases_upload)(zl_ctx* ctx, const void* xy, size_t n, size_t stride, long inf_off, unsigned flags, zl_bases* out) {
    return bases_upload_t<ZL_G>(ctx, xy, n, stride, inf_off, flags, out);
}
template <class G>
static int bases_generate_t(zl_ctx* ctx, const uint64_t* k, size_t n, zl_bases* out) {
    using F = typename G::F;
    if (n >= (1ull << 31)) return ZL_EINVAL;
    void* d_pts = nullptr;
    ZL_HIP(ctx, hipMalloc(&d_pts, std::max<size_t>(n, 1) * sizeof(Affine<F>)));
    if (n) {
        void* d_k;
        int rc;
        const Affine<F>* d_tab = nullptr;
        if ((rc = fb_table_get<G>(ctx, &d_tab))) { (void)hipFree(d_pts); return rc; }  // uses slots 5 / 6 itself: before d_k is bound
        void* d_x;
        if ((rc = zl_scratch_get(ctx, 5, n * 32 + 256 + n * sizeof(XYZZ<F>), &d_k))) { (void)hipFree(d_pts); return rc; }
        d_x = reinterpret_cast<unsigned char*>(d_k) + ((n * 32 + 255) / 256) * 256;
        hipStream_t st = ctx->stream;
        hipError_t e = hipMemcpyAsync(d_k, k, n * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL((k_bases_generate_fb<G>), dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, st, (const uint32_t*)d_k, (uint32_t)n, d_tab, (XYZZ<F>*)d_x);
            e = hipGetLastError();
        }
        if (e == hipSuccess) rc = batch_affine_t<G, 0>(ctx, d_x, n, (Affine<F>*)d_pts, 6, st);
        if (e == hipSuccess && !rc) e = hipStreamSynchronize(st);
        if (rc || e != hipSuccess) { if (!rc) { ctx->last_hip = (int)e; rc = ZL_EHIP; } (void)hipStreamSynchronize(st); (void)hipFree(d_pts); return rc; }
    }
    out->d_pts = d_pts;
    out->n = n;
    {
        const int rc2 = bases_flag_inf_t<G>(ctx, out);  // k_i = 0 (mod r) gives the point at infinity
        if (rc2) { (void)hipFree(d_pts); out->d_pts = nullptr; return rc2; }
    }
    return ZL_OK;
}
int ZL_GNAME(zl_bases_generate)(zl_ctx* ctx, const uint64_t* k, size_t n, zl_bases* out) {
    return bases_generate_t<ZL_G>(ctx, k, n, out);
}
template <class G>
static int bases_download_t(zl_ctx* ctx, const zl_bases& b, size_t first, size_t count, uint64_t* out_xy) {
    using F = typename G::F;
    if (!count) return ZL_OK;
    void* d_out;
    int rc;
    const size_t rec = (size_t)2 * FieldIO<F>::WORDS * 4;
    if ((rc = zl_scratch_get(ctx, 5, count * rec, &d_out))) return rc;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL((k_bases_export<G>), dim3((uint32_t)((count + 127) / 128)), dim3(128), 0, st,
                       reinterpret_cast<const Affine<F>*>(b.d_pts) + first, (uint32_t)count, (uint32_t*)d_out);
    ZL_HIP(ctx, hipGetLastError());
    ZL_HIP(ctx, hipMemcpyAsync(out_xy, d_out, count * rec, hipMemcpyDeviceToHost, st));
    ZL_HIP(ctx, hipStreamSynchronize(st));
    return ZL_OK;
}
int ZL_GNAME(zl_bases_download)(zl_ctx* ctx, const zl_bases& b, size_t first, size_t count, uint64_t* out_xy) {
    return bases_download_t<ZL_G>(ctx, b, first, count, out_xy);
}
