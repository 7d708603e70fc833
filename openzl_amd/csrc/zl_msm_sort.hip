// zl_msm_sort.hip -- steps 1 and 2 of the MSM (zl_msm.hip): signed-digit recoding of the scalars and the (window, bucket) counting sort of
// (point index, sign) entries, for every group alike (the kernels see scalars and digits only).  Replaces the per-window bucket walk of
// ark_ec::msm::VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0; /root/reference/plugins/arkworks/src/groth16.rs:454).
#include "zl_msm_sort.h"

// ------------------------------------------------------------------------------------------------ digits


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

// GLV half-scalars (k_glv_split): a 127-bit magnitude in words 0..3 and the sign in bit 31 of word 7.  The recoders strip the sign off
// the record and fold it into the sign of every digit.
__device__ __forceinline__ uint32_t zl_take_sign(uint32_t& top_word, int glv) {
    if (!glv) return 0u;
    const uint32_t sg = top_word >> 31;
    top_word &= 0x7FFFFFFFu;
    return sg;
}
// MODE 0: histogram; MODE 1: scatter (cursor initialised with the bucket offsets)
__global__ void __launch_bounds__(64) k_msm_zero_words(uint32_t* __restrict__ p, uint32_t n) {
    if (threadIdx.x < n) p[threadIdx.x] = 0u;
}
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
__global__ void __launch_bounds__(256) k_msm_recode(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, int spread_t, int glv, uint16_t* __restrict__ digits,
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
__global__ void __launch_bounds__(1024) k_msm_hist_lds(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t per_slice, uint32_t NB,
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
__global__ void __launch_bounds__(256) k_msm_slice_prefix(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ tot) {
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
__global__ void __launch_bounds__(1024) k_msm_scatter_range(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t RB,
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
__global__ void __launch_bounds__(256) k_msm_recode_wide(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, uint32_t gw, int spread_t, int glv,
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
__global__ void __launch_bounds__(256) k_msm_part_hist(const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W, uint32_t G, uint32_t per_slice,
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
__global__ void __launch_bounds__(256) k_msm_sub_hist(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_off, uint32_t G,
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
__global__ void __launch_bounds__(256) k_msm_part_scatter_st(const uint16_t* __restrict__ lo16, const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W,
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
__global__ void __launch_bounds__(256) k_msm_sub_scatter_st(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_idx,
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
__global__ void __launch_bounds__(256) k_msm_fine_hist(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ sub_off, uint32_t SG,
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
__global__ void __launch_bounds__(1024) k_msm_fine_sort(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
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
__global__ void __launch_bounds__(1024) k_msm_fine_sort_big(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
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

// ------------------------------------------------------------------------------------------------ scan
// exclusive scan of `count` u32 values, 3 launches; out[count] = total
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ block_sums) {
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
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t* __restrict__ total_out,
                                                          const uint32_t* __restrict__ flag_in /* copied to total_out[1] */) {
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
// Small bucket counts (NB <= 14336): slice prefix + the three scan launches as ONE block of 1024 lanes.
// counts[slice][bucket] -> in-place exclusive prefix over slices; offsets[b] = cursor[b] = exclusive prefix over buckets; offsets[NB] = total,
// offsets[NB + 1] = *flag_in (as k_scan_top).  Four dependent launches of a few microseconds each are what a small job's sort phase consists of.
// Round 6: the slice walk runs with consecutive lanes on consecutive buckets (coalesced) and leaves the bucket totals in LDS, where the scan's 16
// buckets per lane are read -- the first form gave every lane 16 consecutive buckets in GLOBAL memory (64-byte lane stride, 16 x nslices dependent
// read-modify-writes per lane): 100 us at 12 288 buckets x 8 slices, a fifth of a 2^14-point MSM's device time (profiles/r06_timeline_msm_2_14_glv.txt).
// LDS: NB + NB / 16 words (bucket b at b + b / 16: the stride of a lane's 16 buckets becomes 17 words, no bank conflicts).
__global__ void __launch_bounds__(1024) k_msm_prefix_small(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ cursor, const uint32_t* __restrict__ flag_in) {
    ZL_SIDE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* tot = reinterpret_cast<uint32_t*>(smem);
    __shared__ uint32_t sh[1024];
    for (uint32_t b = threadIdx.x; b < NB; b += 1024) {
        uint32_t run = 0;
        for (uint32_t sl = 0; sl < nslices; sl++) {
            const uint32_t x = counts[(size_t)sl * NB + b];
            counts[(size_t)sl * NB + b] = run;
            run += x;
        }
        tot[b + (b >> 4)] = run;
    }
    __syncthreads();
    const uint32_t base = threadIdx.x * 16;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (base + k < NB) s += tot[threadIdx.x * 17 + k];
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
    for (int k = 0; k < 16; k++)
        if (base + k < NB) {
            const uint32_t x = tot[threadIdx.x * 17 + k];
            tot[threadIdx.x * 17 + k] = run;
            run += x;
        }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < NB; b += 1024) {
        const uint32_t o = tot[b + (b >> 4)];
        offsets[b] = o;
        cursor[b] = o;
    }
    if (threadIdx.x == 1023) {
        offsets[NB] = sh[1023];
        if (flag_in) offsets[NB + 1] = *flag_in;
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* __restrict__ in, uint32_t count, const uint32_t* __restrict__ block_sums,
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

// the two modes of the global-atomics sort (plain windows of 21 bits and more): 0 = histogram, 1 = scatter
template __global__ void k_msm_digits<0>(const uint32_t*, uint32_t, int, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, const uint8_t*, int, uint32_t*);
template __global__ void k_msm_digits<1>(const uint32_t*, uint32_t, int, int, uint32_t*, uint32_t*, uint32_t*, uint32_t*, const uint8_t*, int, uint32_t*);
