// zl_msm_job.h -- one MSM as a plan over device buffers: window choice, buffer layout, and the launch sequences of its three phases
// (sort | accumulate | tail) plus the host Horner.  The drivers in zl_msm.hip run one job, or pipeline several on three streams.
#pragma once
#include <chrono>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "zl_ctx.h"
#include "zl_pool.h"
#include "zl_msm_sort.h"
#include "zl_msm_endo.h"
#include "zl_msm_accumulate.h"
#include "zl_msm_reduce.h"

// ------------------------------------------------------------------------------------------------ host driver
// developer tuning knobs (profiling sweeps only; unset in production): ZL_TUNE_CHUNK, ZL_TUNE_SEG, ZL_TUNE_FS, ZL_TUNE_RANGES (zl_tune, zl_ctx.h)
static int zl_pick_window(size_t n, int sc_bits, bool wide16 = false /* c = 16 also runs the three-level sort (GLV jobs) */) {
    // cost in accumulated entries: n per window (+10 % for c <= 16: the one-level LDS counting sort streams every window's digits once per
    // bucket range and is the slower sort at large n) + ~4.5 per bucket (merge of cut buckets, level-0 running sums, tree).  Fitted on
    // single-call times at 2^20 .. 2^24, both curves (profiles/r02_msm_sweep_plain.log, r02_msm_sweep_bn254.log), refitted in round 6 (below).  c = 17..20 run the three-level sort over W bucket sets (<= 255 sort groups).
    // (Round 6: 5.7 -> 5.2 with the merge by chunk boundaries, then 4.5: pipelined sweeps of every window at 2^21 .. 2^23 (profiles/r06_window_sweep_2_21_23.log) have 18 ahead of 17 at
    // 2^22 by 1.7 % and 19 ahead of 18 at 2^23 by 4 % -- c = 18 is 15 windows with a nearly empty top one, c = 19 is 14 -- and one constant below 4.6 fits all of 2^20: 16, 2^21: 17, 2^22: 18,
    // 2^23: 19, 2^24: 20.)
    // Split scalars (sc_bits 127 / 64) keep 5.2: at 4.5 BN254's 2^18 moves to c = 16, which is faster alone and 7 % slower inside a proof (zl_pick_window_half below).
    const double per_bucket = (double)zl_tune("ZL_TUNE_BUCKET_COST_X10", sc_bits > 200 ? 45 : 52) / 10.0;
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

// The 127-bit half-scalars of an endomorphism split on G1 (2 n entries per window, half the windows): the cost model above is fitted on machine-filling sizes and misses
// what decides a latency-bound MSM (which sort runs, chunk length, four-lane or one-lane kernels): at 2^18 points it picks c = 15 where c = 16 is 13 % faster, at 2^13 / 2^14
// c = 10 / 11 where 11 / 12 are 8-10 % faster.  Measured best window by size, entries only where the gain is clear and survives inside a proof (profiles/r06_glv_window_sweep.log: every c in 9 .. 17
// at 2^11 .. 2^19 points); 0 = outside the table, the model decides.
static int zl_pick_window_half(size_t n_points) {
    if (n_points < 2) return 0;
    int lg = 63 - __builtin_clzll((unsigned long long)n_points);
    if ((double)n_points >= 1.4142 * (double)((size_t)1 << lg)) lg++;  // nearest power of two
    static const signed char best[9] = {9, 0, 11, 12, 0, 0, 0, 16, 0};  // 2^11 .. 2^19 points; 0 = the model (ties as single calls; c = 12 at 2^16 made a 57 000-constraint proof 8 % slower)
    return lg >= 11 && lg <= 19 ? best[lg - 11] : 0;
}
// ... and the 64-bit quarter-scalars of the G2 split (4 n entries per window): the model's 11 at 2^13 / 2^14 where 12 / 13 are 8-10 % faster, 17 at 2^18 where 14 is 5 % faster
// (profiles/r06_half_table_ab.log, "G2 window sweep": every c in 8 .. 16 at 2^10 .. 2^18 points)
static int zl_pick_window_quarter(size_t n_points) {
    if (n_points < 2) return 0;
    int lg = 63 - __builtin_clzll((unsigned long long)n_points);
    if ((double)n_points >= 1.4142 * (double)((size_t)1 << lg)) lg++;
    static const signed char best[9] = {9, 0, 11, 12, 13, 13, 13, 13, 14};  // 2^10 .. 2^18 points (0: not measured, the model)
    return lg >= 10 && lg <= 18 ? best[lg - 10] : 0;
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
    // result buffer (pinned when pipelined): SETS * roots_per_set root channels, the scalar-1 sum, then the two status words k_msm_ones leaves behind it
    void set_host_buffer(X* buf) {
        hw = buf;
        hE = reinterpret_cast<uint32_t*>(buf + (size_t)SETS * roots_per_set + 1);
    }

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
        // BN254 G1 (round 6: two-dimensional split, k_glv_split_lattice): its host Horner is cheap (254 doublings at 0.3 us) against the split's extra device work (2 n
        // sort records, one more kernel), so between 2^13 and 2^17 points the plain windows are 3-4 % faster as a single call and 8 % pipelined -- BASELINE config 1's size
        // among them -- while 2^10 (-15 %), 2^17 (-14 %) and 2^19 (-9 %) gain (profiles/r06_small_knobs2.log, r06_bn_glv_ab.log).  BLS12-381 never loses (same logs).
        if constexpr (G::GLV && G::ENDO_K == 2) {
            if (G::GLVP::LATTICE && n_ >= ((size_t)1 << 13) && n_ < ((size_t)1 << 17)) try_glv = false;
        }
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
        if (!pre && ctx->msm_c <= 0 && glv && zl_tune("ZL_TUNE_HALF_TABLE", 1)) {
            int h = 0;
            if constexpr (G::ENDO_K == 2) {
                // (BN254: c = 16 at 2^18 is 11 % faster as a single call and 7 % SLOWER inside a 240 000-constraint proof, where four such MSMs and the witness map share
                // the machine -- the table is BLS12-381's, profiles/r06_tables_mid_ab.log)
                if (!G::GLVP::LATTICE) h = zl_pick_window_half(n_);
            } else {
                h = zl_pick_window_quarter(n_);
            }
            if (h) c = h;
        }
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
        // ... but never so short that an average bucket is cut into more than ~4 chunks: beyond ZL_BIG_SPAN_SMALL = 8 chunks a bucket leaves the lane-serial fold of
        // k_msm_merge for one WORKGROUP per bucket (k_msm_merge_big), which is meant for the few heavy buckets of a skewed input, not for all of them.  Round 6 found
        // the case in a sweep: a 2^14-point G2 MSM (GLS: 65 536 quarter-scalars, c = 11, 64 entries per bucket, 8-entry chunks) spent 3 of its 3.5 ms there --
        // slower than the 2^16-point MSM (1.4 ms).
        // (Half-scalars of an endomorphism split fill only part of their top window's range -- |k_i| <= 0.67 * 2^127 on BLS12-381, 0.43 * 2^127 on BN254 -- so that
        // window's buckets are 1.5-2.3x as dense as the average: counted double.  A 2^14-point G1 MSM, 32 entries per bucket on average and 8-entry chunks, sent a
        // handful of its top window's buckets to k_msm_merge_big: 82 us of a 0.56-ms device chain, profiles/r06_timeline_msm_2_14_glv.txt.)
        while (ZL_CHUNK < (uint32_t)ZL_CHUNK_MAX && (glv && G::ENDO_K == 2 ? 2u : 1u) * (maxE / std::max<uint64_t>(1, (uint64_t)SETS * H)) > (uint64_t)4 * ZL_CHUNK) ZL_CHUNK <<= 1;
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
        hw_own.assign((size_t)SETS * roots_per_set + 2, X::inf());  // (+ one element of room for the two status words behind the result)
        set_host_buffer(hw_own.data());
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
        d_ones_parts = d_sets + root_elems + 2;  // (one element of room behind the scalar-1 sum: the two status words k_msm_ones writes there)
        d_giant_tmp = d_ones_parts + ZL_ONES_BLOCKS;
        if (glv) {
            // The images depend on the bases only: kept with the handle (one range per handle; another range of the same handle falls back to
            // the per-call scratch copy below).  k_gls_psi is 50 us of latency in front of the G2 MSM of every small proof, k_glv_phi 12-100 us.
            const size_t endo_bytes = (size_t)(endo_k - 1) * n_real * sizeof(Affine<F>);
            const zl_bases& bs = *bsp;
            std::lock_guard<std::mutex> cache_lk(zl_bases_cache_mutex());  // the handle may be shared by the lanes of a forked ctx
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
        int rc;
        // the bucket counters are written in full by the LDS path (k_msm_slice_prefix) and by the wide path (k_msm_fine_hist); only the
        // global-atomics sort counts into them
        if (!wide && c > 16) ZL_HIP(ctx, hipMemsetAsync(d_counts, 0, (size_t)(NB + 1) * 4, st));
        hipLaunchKernelGGL(k_msm_zero_words, dim3(1), dim3(64), 0, st, d_big_count, 8u);  // big, ones, giant counts, bad-scalar flag; [4..5]: oversized sub-group queue head (u64); [6]: k_msm_ones' ticket
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
                if constexpr (G::GLVP::LATTICE)
                    hipLaunchKernelGGL((k_glv_split_lattice<typename G::GLVP>), dim3((uint32_t)((n_real + 255) / 256)), dim3(256), 0, st, sc, (uint32_t)n_real, d_inf, d_vs, (int)G::SC_BITS, d_bad_scalar);
                else
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
        if (wide) rc = sort_wide(ctx, st, sc_eff, inf_eff, bad_eff, nblk, glv_i);
        else if (c <= 16) rc = sort_lds(ctx, st, sc_eff, inf_eff, bad_eff, nblk, glv_i);
        else rc = sort_atomics(ctx, st, nblk);
        if (rc) return rc;
        ZL_HIP(ctx, hipGetLastError());
        return ZL_OK;
    }
    // ---- three-level counting sort over (window, bucket) ids: the merged set of a table, or W sets of plain wide windows (c = 17 .. 20; GLS quarter-scalars at c = 16)
    int sort_wide(zl_ctx* ctx, hipStream_t st, const uint32_t* sc_eff, const uint8_t* inf_eff, uint32_t* bad_eff, uint32_t nblk, int glv_i) {
        const zl_bases& bs = *bsp;
        int rc;
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
        return ZL_OK;
    }
    // ---- LDS counting sort (c <= 16): recode once (u16 digits), per-(slice, window) LDS histograms, slice prefix, scan, range-owned scatter
    int sort_lds(zl_ctx* ctx, hipStream_t st, const uint32_t* sc_eff, const uint8_t* inf_eff, uint32_t* bad_eff, uint32_t nblk, int glv_i) {
        int rc;
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
        if (NB <= 14336 && nslices <= 64) {  // (14 336 buckets: 59.6 KB of LDS for the totals beside the 4 KB of the scan)
            hipLaunchKernelGGL(k_msm_prefix_small, dim3(1), dim3(1024), ((size_t)NB + NB / 16 + 16) * 4, st, d_slice_counts, NB, nslices, d_offsets, d_cursor, (const uint32_t*)d_bad_scalar);
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
        const uint32_t parts = 1;
        hipLaunchKernelGGL(k_msm_scatter_range, dim3(8 * ((W + 7) / 8), ranges, parts), dim3(1024), (size_t)RB * 4, st, d_digits, (uint32_t)n, H, RB, d_offsets, d_entries,
                           (const uint32_t*)d_slice_counts, NB, nslices, per_slice, parts, (uint32_t)W);
        return ZL_OK;
    }
    // ---- wide windows without a table beyond 255 sort groups (forced plain c >= 21): histogram / scatter with global atomics
    int sort_atomics(zl_ctx* ctx, hipStream_t st, uint32_t nblk) {
        // wide windows without a table: histogram / scatter with global atomics
        hipLaunchKernelGGL((k_msm_digits<0>), dim3(nblk), dim3(256), 0, st, sc, (uint32_t)n, c, W, d_counts, (uint32_t*)nullptr, d_ones_list, d_ones_count, d_inf, (int)G::SC_BITS, d_bad_scalar);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums);
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, d_block_sums, scan_blocks, d_offsets + NB, (const uint32_t*)d_bad_scalar);
        hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, st, d_counts, NB, d_block_sums, d_offsets, d_cursor);
        hipLaunchKernelGGL((k_msm_digits<1>), dim3(nblk), dim3(256), 0, st, sc, (uint32_t)n, c, W, d_cursor, d_entries, d_ones_list, d_ones_count, d_inf, (int)G::SC_BITS, d_bad_scalar);
        (void)ctx;
        return ZL_OK;
    }
    static constexpr bool pair_ok() { return G::COORDS == 2 && !std::is_void<typename PairBase<F>::type>::value; }  // an Fq2 group on 28-bit limbs: the lane-pair kernels exist
    int accumulate(zl_ctx* ctx, hipStream_t st) {
        if (pair_ok() && zl_tune("ZL_TUNE_G2_OCTET", 1) && nchunks <= (uint64_t)zl_tune("ZL_TUNE_QUAD_ACC_CHUNKS", 24576))  // Fq2 groups: eight lanes per chunk (zl_fq2pair.h)
            hipLaunchKernelGGL((k_msm_accumulate_pair<G, true>), dim3((nchunks + 7) / 8), dim3(64), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                               glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        else if (nchunks <= (uint64_t)zl_tune("ZL_TUNE_QUAD_ACC_CHUNKS", 24576))  // four lanes per chunk while that is at most ~1.5 waves per SIMD (round 6: 49 152 -> 24 576; at three waves per SIMD the one-lane kernel is faster: 94 against 131 us at 2^14, profiles/r06_small_knobs2.log)
            hipLaunchKernelGGL((k_msm_accumulate_quad<G>), dim3((4 * nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK), dim3(ZL_ACC_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                               glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        else if (pair_ok() && zl_tune("ZL_TUNE_G2_PAIR", 1))  // Fq2 groups: two lanes per chunk, two waves per SIMD (zl_fq2pair.h)
            hipLaunchKernelGGL((k_msm_accumulate_pair<G, false>), dim3((nchunks + 31) / 32), dim3(64), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                               glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
#ifdef ZL_MEASURE
        else if (G::COORDS == 1 && ctx->acc_clk && (size_t)((nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK) * 32 <= ctx->acc_clk_cap) {  // armed by the measurement hook zl_test_acc_clock only
            ctx->acc_clk_waves = (nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK;
            hipLaunchKernelGGL((k_msm_accumulate_clk<G>), dim3((nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK), dim3(ZL_ACC_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                               glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu, (unsigned long long*)ctx->acc_clk,
                               zl_tune("ZL_TUNE_ACC_CLK_IDX_BITS", 31) >= 31 ? 0x7fffffffu : ((1u << zl_tune("ZL_TUNE_ACC_CLK_IDX_BITS", 31)) - 1u));
        } else
#else
        else
#endif
        hipLaunchKernelGGL((k_msm_accumulate<G>), dim3((nchunks + ZL_ACC_BLOCK - 1) / ZL_ACC_BLOCK), dim3(ZL_ACC_BLOCK), 0, st, d_entries, d_offsets, NB, d_bases, d_buckets, d_partials, ZL_CHUNK,
                           glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu);
        ZL_HIP(ctx, hipGetLastError());
        return ZL_OK;
    }
    int tail(zl_ctx* ctx, hipStream_t st) {
        // four lanes per group operation (zl_quad.h) in every tail launch that does not fill the machine
        const uint32_t quad_max = (uint32_t)zl_tune("ZL_TUNE_QUAD_LANES", 65536);
        const bool pair_tails = pair_ok() && zl_tune("ZL_TUNE_G2_PAIR_TAILS", 1), octet = pair_ok() && zl_tune("ZL_TUNE_G2_OCTET", 1);  // Fq2 groups: two lanes per item where a launch fills the machine
        const bool by_cuts = NB > quad_max && nchunks > 1 && zl_tune("ZL_TUNE_MERGE_CUTS", 1);  // one lane (pair) per chunk boundary: every surviving lane folds one bucket
        if (by_cuts) {
            hipLaunchKernelGGL((k_msm_fill_empty<G>), dim3((NB + 255) / 256), dim3(256), 0, st, d_offsets, NB, d_buckets);
            if (pair_tails)
                hipLaunchKernelGGL((k_msm_merge_cuts_pair<G>), dim3((nchunks + 31) / 32), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span, nchunks);
            else
                hipLaunchKernelGGL((k_msm_merge_cuts<G>), dim3((nchunks + 63) / 64), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span, nchunks);
        } else if (pair_tails && NB > quad_max)
            hipLaunchKernelGGL((k_msm_merge_pair<G, false>), dim3((NB + 31) / 32), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        else if (octet && NB <= quad_max)
            hipLaunchKernelGGL((k_msm_merge_pair<G, true>), dim3((NB + 7) / 8), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        else if (NB <= quad_max)
            hipLaunchKernelGGL((k_msm_merge<G, true>), dim3((4 * NB + 63) / 64), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        else
        hipLaunchKernelGGL((k_msm_merge<G>), dim3((NB + 63) / 64), dim3(64), 0, st, d_offsets, NB, d_buckets, d_partials, d_big_list, d_big_count, d_giant_list, d_giant_count, ZL_CHUNK, big_span);
        // a bucket cut into more than big_span (ZL_GIANT_SPAN) chunks needs that many chunks to exist: small jobs skip the launches (three of the ~22 of a small MSM's chain)
        const bool may_big = nchunks > big_span, may_giant = nchunks > (uint32_t)ZL_GIANT_SPAN;
        if (pair_tails && zl_tune("ZL_TUNE_G2_PAIR_BLOCKS", 1)) {  // Fq2 groups: the block-tree kernels of the heavy buckets on lane pairs
            if (may_big) hipLaunchKernelGGL((k_msm_merge_big_pair<G>), dim3(std::min<uint32_t>(max_big, 1024)), dim3(2 * TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_offsets, d_buckets,
                               d_partials, d_big_list, d_big_count, ZL_CHUNK);
            if (may_giant) hipLaunchKernelGGL((k_msm_merge_giant_pair<G>), dim3(std::min<uint32_t>(max_giant, 16) * ZL_GIANT_PARTS), dim3(2 * TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st,
                               d_offsets, d_giant_tmp, d_partials, d_giant_list, d_giant_count, ZL_CHUNK);
            if (may_giant) hipLaunchKernelGGL((k_msm_merge_giant2_pair<G>), dim3(std::min<uint32_t>(max_giant, 64)), dim3(2 * ZL_GIANT_PARTS), 0, st, d_buckets, d_giant_tmp, d_giant_list, d_giant_count);
        } else {
            if (may_big) hipLaunchKernelGGL((k_msm_merge_big<G>), dim3(std::min<uint32_t>(max_big, 1024)), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_offsets, d_buckets,
                           d_partials, d_big_list, d_big_count, ZL_CHUNK);
            if (may_giant) hipLaunchKernelGGL((k_msm_merge_giant<G>), dim3(std::min<uint32_t>(max_giant, 16) * ZL_GIANT_PARTS), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st,
                           d_offsets, d_giant_tmp, d_partials, d_giant_list, d_giant_count, ZL_CHUNK);
            if (may_giant) hipLaunchKernelGGL((k_msm_merge_giant2<G>), dim3((max_giant + 63) / 64), dim3(64), 0, st, d_buckets, d_giant_tmp, d_giant_list, d_giant_count);
        }
        // scalar-1 bases: window-0 table entries are the bases themselves
        hipLaunchKernelGGL((k_msm_ones<G>), dim3(ZL_ONES_BLOCKS), dim3(TreeLanes<G>::N), TreeLanes<G>::N * sizeof(X), st, d_ones_list, d_ones_count,
                           pre ? d_bases + first : d_bases, d_ones_parts, glv ? d_phi : d_bases, glv ? (uint32_t)n_real : 0xFFFFFFFFu,
                           d_sets + (size_t)SETS * roots_per_set, d_big_count + 6, (const uint32_t*)(d_offsets + NB));  // (+ the final sum and the two status words: see the kernel)
        {
            const uint32_t fset = spread_t >= 0 ? (uint32_t)(W - 1) : 0xFFFFFFFFu, flog = (uint32_t)std::max(spread_t, 0);
            const uint32_t leaves = SETS * red_blocks;
            X* cur = red_levels == 0 ? d_sets : d_segs;
            if (pair_tails && leaves > quad_max)
                hipLaunchKernelGGL((k_msm_reduce_level0_pair<G, false>), dim3((leaves + 31) / 32), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            else if (octet && leaves <= quad_max)
                hipLaunchKernelGGL((k_msm_reduce_level0_pair<G, true>), dim3((leaves + 7) / 8), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            else if (leaves <= quad_max)
                hipLaunchKernelGGL((k_msm_reduce_level0<G, true>), dim3((4 * leaves + 63) / 64), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            else
            hipLaunchKernelGGL((k_msm_reduce_level0<G>), dim3((leaves + 63) / 64), dim3(64), 0, st, d_buckets, H, red_g0, red_blocks, leaves, fset, flog, cur);
            for (uint32_t lv = 1; lv <= red_levels; lv++) {
                const uint32_t nodes = red_blocks >> lv, lanes = SETS * nodes * (lv + 2);
                X* nxt = lv == red_levels ? d_sets : ((lv & 1) ? d_stage1 : d_segs);
                if (pair_tails && lanes > quad_max)
                    hipLaunchKernelGGL((k_msm_reduce_tree_pair<G, false>), dim3((lanes + 31) / 32), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                else if (octet && lanes <= quad_max)
                    hipLaunchKernelGGL((k_msm_reduce_tree_pair<G, true>), dim3((lanes + 7) / 8), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                else if (lanes <= quad_max)
                    hipLaunchKernelGGL((k_msm_reduce_tree<G, true>), dim3((4 * lanes + 63) / 64), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                else
                hipLaunchKernelGGL((k_msm_reduce_tree<G>), dim3((lanes + 63) / 64), dim3(64), 0, st, cur, nxt, lv, nodes, lanes);
                cur = nxt;
            }
        }
        ZL_HIP(ctx, hipGetLastError());
        ZL_HIP(ctx, hipMemcpyAsync(hw, d_sets, sizeof(X) * ((size_t)SETS * roots_per_set + 1) + 8, hipMemcpyDeviceToHost, st));  // root channels, scalar-1 sum, the two status words
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
    X finish(bool parallel = true, double* values_us = nullptr) const {  // values_us (developer trace): time of stage 1
        std::vector<X> V(SETS);
        const auto t0 = std::chrono::steady_clock::now();
        if (parallel && SETS >= 4) zl_pool_get().parallel_for(SETS, [&](size_t w) { V[w] = window_value((int)w); });
        else for (uint32_t w = 0; w < SETS; w++) V[w] = window_value((int)w);
        if (values_us) *values_us = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1e3;
        X total = V[SETS - 1];
        for (int w = (int)SETS - 2; w >= 0; w--) {
            zl::dbl_n(total, c);  // c doublings in Jacobian coordinates
            zl::add_full(total, V[w]);
        }
        zl::add_full(total, hw[(size_t)SETS * roots_per_set]);
        return total;
    }
};
