// zl_ctx.h -- internal context shared by the MSM / NTT translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <vector>
#include "../../include/zl_backend_ext.h"
#include "zl_curve.h"
#include "zl_pool.h"

#define ZL_HIP(ctx, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) {                             \
            (ctx)->last_hip = (int)e_;                      \
            return e_ == hipErrorOutOfMemory ? ZL_ENOMEM : ZL_EHIP; \
        }                                                   \
    } while (0)

// ---- group configurations ---------------------------------------------------------------------------------
struct BlsG1 {
    static constexpr int ID = 0;
    using FqP = BLS12_381_Fq;
    using FrP = BLS12_381_Fr;
    using F = Fp28<BLS12_381_Fq28, BLS12_381_Fq>;  // 14 x 28-bit lazily reduced limbs (zl_field28.h): +30 % multiplier throughput
    using C = BLS12_381_G1;
    static constexpr int SC_BITS = 255;
    static constexpr int FQ64 = 6;     // u64 limbs per Fq element in the ABI layouts
    static constexpr int COORDS = 1;   // Fq elements per coordinate
    static constexpr bool GLV = true;  // plain MSMs split every scalar with the endomorphism (x, y) -> (beta x, y) = [lambda](x, y)
    static constexpr int ENDO_K = 2;   // ... into two 127-bit halves
    using GLVP = BLS12_381_GLV;
    ZL_HD static F glv_beta() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = GLVP::beta(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F gen_x() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::gx(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F gen_y() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::gy(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F coeff_b() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::b(i); return FieldIO<F>::load_mont32(w); }
};
struct BnG1 {
    static constexpr int ID = 1;
    using FqP = BN254_Fq;
    using FrP = BN254_Fr;
#ifdef ZL_BN_FIELD32
    using F = Fp<FqP>;  // rounds 1-3: 8 x 32-bit carry-chain limbs (developer A/B switch)
#else
    using F = Fp28<BN254_Fq28, BN254_Fq>;  // round 4: 10 x 28-bit lazily reduced limbs like BLS12-381 (zl_field28.h)
#endif
    using C = BN254_G1;
    static constexpr int SC_BITS = 254;
    static constexpr int FQ64 = 4;
    static constexpr int COORDS = 1;
    static constexpr bool GLV = true;  // round 6: the same endomorphism split as BLS12-381 G1, with the two-dimensional lattice decomposition (BN254_GLV, k_glv_split_lattice)
    static constexpr int ENDO_K = 2;
    using GLVP = BN254_GLV;
    ZL_HD static F glv_beta() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = GLVP::beta(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F gen_x() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::gx(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F gen_y() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::gy(i); return FieldIO<F>::load_mont32(w); }
    ZL_HD static F coeff_b() { uint32_t w[FqP::N]; for (int i = 0; i < FqP::N; i++) w[i] = C::b(i); return FieldIO<F>::load_mont32(w); }
};
template <class C2, class FqP_, class FrP_, class F_, int SCB, int FQ64_, int ID_>
struct G2Cfg {
    static constexpr int ID = ID_;
    using FqP = FqP_;
    using FrP = FrP_;
    using F = F_;
    static constexpr int SC_BITS = SCB;
    static constexpr int FQ64 = FQ64_;
    static constexpr int COORDS = 2;
    // BLS12-381 G2 (ID 2): the untwist-Frobenius-twist endomorphism psi acts as [z] = [-|z|], r < |z|^4: a scalar is four base-|z| digits of 64 bits
    // over P, psi(P), psi^2(P), psi^3(P) with alternating signs (GLS).  BN254 G2 keeps plain 254-bit windows.
    static constexpr bool GLV = ID_ == 2;
    static constexpr int ENDO_K = ID_ == 2 ? 4 : 1;
    using GLVP = BLS12_381_GLS;
    ZL_HD static F psi_x() { return mk(GLVP::psi_x0, GLVP::psi_x1); }
    ZL_HD static F psi_y() { return mk(GLVP::psi_y0, GLVP::psi_y1); }
    ZL_HD static F mk(uint32_t (*f0)(int), uint32_t (*f1)(int)) {  // two components as arkworks' Montgomery words
        uint32_t w[2 * FqP::N];
        for (int i = 0; i < FqP::N; i++) { w[i] = f0(i); w[FqP::N + i] = f1(i); }
        return FieldIO<F>::load_mont32(w);
    }
    ZL_HD static F gen_x() { return mk(C2::gx0, C2::gx1); }
    ZL_HD static F gen_y() { return mk(C2::gy0, C2::gy1); }
    ZL_HD static F coeff_b() { return mk(C2::b0, C2::b1); }
};
// G2 on the lazily reduced 28-bit fields (Fq2 products as dual scans, zl_field28.h): 14 limbs per component for BLS12-381, 10 for BN254
using BlsG2 = G2Cfg<BLS12_381_G2, BLS12_381_Fq, BLS12_381_Fr, Fp2L<Fp28<BLS12_381_Fq28, BLS12_381_Fq>>, 255, 6, 2>;
#ifdef ZL_BN_FIELD32
using BnG2 = G2Cfg<BN254_G2, BN254_Fq, BN254_Fr, Fp2<BN254_Fq>, 254, 4, 3>;
#else
using BnG2 = G2Cfg<BN254_G2, BN254_Fq, BN254_Fr, Fp2L<Fp28<BN254_Fq28, BN254_Fq>>, 254, 4, 3>;  // round 4: like BLS12-381 G2, on 10 limbs
#endif

// ---- context ----------------------------------------------------------------------------------------------
struct zl_bases {
    void* d_pts = nullptr;  // Affine<F>[n], Montgomery
    void* d_table = nullptr;  // optional Affine<F>[W][n]: 2^(c w) P_i (zl_bases_precompute)
    int precomp_c = 0;
    void* d_inf = nullptr;   // optional uint8[n]: 1 = the point at infinity (present only when the handle holds at least one)
    size_t n_inf = 0;
    size_t n = 0;
    int curve = 0, group = 0;
    // endomorphism images of points [endo_first, endo_first + endo_n) (GLV: phi(P); GLS: psi, psi^2, psi^3), built by the first small MSM over
    // that range and kept with the handle: a proving key's queries are always used with the same range
    mutable void* d_endo = nullptr;
    mutable size_t endo_first = 0, endo_n = 0;
    mutable int endo_k = 0;
    mutable std::vector<uint64_t> first_xy;  // canonical affine words of point 0, fetched once (Groth16: the z_0 = 1 term of a / b queries)
    // Groth16 per-key state hung on the l_query handle of a proving key (zl_groth16.hip: host tables of the key's fixed points, the folded C query of small
    // proofs); built on the first proof over the key under zl_bases_cache_mutex, released with the handle
    mutable std::shared_ptr<void> g16_cache;
};
struct zl_scratch {
    void* p = nullptr;
    size_t cap = 0;
};
struct zl_twiddles {
    void* d_lo = nullptr;  // w^i, i < 2^lo_bits
    void* d_hi = nullptr;  // w^(i << lo_bits)
    void* d_small = nullptr;  // per-radix tables
    void* d_small_limbs = nullptr;  // the per-radix tables as limbs (read by the lazy passes' butterflies)
    void* d_last = nullptr;   // combined inter-factor twiddles of the last pass, one per element (multi-pass sizes, built on first use)
    void* d_row[4] = {nullptr, nullptr, nullptr, nullptr};  // per-row twiddles of a MIDDLE pass (index: pass - 1), one row per value of the tile's high index; lazy passes only
    size_t last_bytes = 0;    // ... its size, and when it was last used: the tables of one ctx share a byte budget (ZL_TUNE_NTT_LAST_MB), least recently used first out
    uint64_t last_used = 0;
    unsigned lo_bits = 0;
};
struct zl_r1cs_dev {  // device-resident R1CS matrices (CSR; coefficients in Montgomery form)
    void* d_base = nullptr;
    size_t off_ptr[3] = {0, 0, 0}, off_col[3] = {0, 0, 0}, off_val[3] = {0, 0, 0};
    uint32_t n_constraints = 0, n_instance = 0, n_witness = 0;
    size_t nnz[3] = {0, 0, 0};  // non-zeros of A, B, C (the witness map picks the row-parallel product for matrices with long rows)
    int curve = 0;
};
struct zl_ctx {
    // A FORK (zl_ctx_fork) is a second prover lane on the parent's device: its own streams, scratch, events, workers and twiddle tables, and READ access to the
    // parent's device-resident objects (bases with their tables, R1CS matrices) through zl_find_bases / zl_find_r1cs -- so N host threads prove side by side
    // over ONE copy of a proving key, like N threads sharing the reference's `&ProvingContext` (groth16.rs:445-457 takes it by shared reference).
    zl_ctx* parent = nullptr;
    std::atomic<int> forks{0};  // live forks of this ctx: its handles cannot be freed while any exists
    mutable std::shared_mutex maps_mu;  // guards the two handle maps below: lookups from the lanes (shared) against uploads (exclusive); entries are node-stable
    std::vector<zl_ctx*> fork_list;  // the live forks (under maps_mu): a parent destroyed first orphans them (parent = nullptr: their lookups of its handles then fail with ZL_EHANDLE instead of reading freed memory)
    int fork_seq = 0;           // forks number their own handles from (seq << 48) | 1: no value of theirs collides with one of the parent's
    int device = 0;
    int cu_count = 0;                  // compute units of the device (grid of the persistent accumulation kernel)
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // created by the ctx
    int last_hip = 0;
    int msm_c = 0;
    int timing_on = 0;
    zl_timing timing{};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::map<uint64_t, zl_bases> bases;
    std::map<uint64_t, zl_r1cs_dev> r1cs;
    uint64_t next_handle = 1;
    zl_scratch scratch[40];  // 0..9: first buffer set + shared; 10..13 / 14..17: second / third MSM buffer set (pipelined batches); 18, 20..22: endomorphism images of the bases (GLV); 4, 19, 23: tail buffers of the three sets
    std::map<uint64_t, zl_twiddles> twiddles;  // key: curve<<16 | log_n<<1 | inverse
    size_t ntt_last_bytes = 0;   // bytes held by the d_last tables of this ctx
    // events of the job pipelines, created once and reused by every call: a hipEventCreate / hipEventDestroy pair costs ~50 us of host time, and a
    // four-job pipeline used 14 of them per call -- 0.8 ms at the END of every Groth16 proof (round 4 host trace, profiles/r04_g16_share_ab.log)
    std::vector<hipEvent_t> ev_pool[2];  // [0] hipEventDisableTiming, [1] timing
    // A ctx is SINGLE-CALLER (include/zl_backend.h): the pools above, the scratch slots and the streams belong to the one call in flight.  The job pipeline
    // holds this flag while it uses pool events (zl_ctx_events returns a pointer into the vector: a second, concurrent pipeline on the same ctx would alias
    // the events and a growing pool would move them -- ADVICE r4); a second caller gets ZL_EINVAL instead of a race.  Concurrency = one ctx per thread
    // (Groth16 runs its G2 MSM and witness map on ctx->aux / aux2 for that reason).
    std::atomic<int> pipeline_busy{0};
    uint64_t ntt_clock = 0;
    hipStream_t stream_sort = nullptr;  // pipelined MSM batches: sort | accumulate (ctx->stream) | tail
    hipStream_t stream_lane[4] = {nullptr, nullptr, nullptr, nullptr};  // batches of SMALL MSMs: every job runs sort, accumulation and tail on the stream of its buffer set, the jobs side by side
    hipStream_t stream_tail[3] = {nullptr, nullptr, nullptr};  // one tail stream per buffer set: the tails of consecutive small jobs run side by side
    hipStream_t stream_copy = nullptr;  // zl_msm with host scalars: chunked H2D copies that run under the MSMs of the earlier chunks
    void* pinned = nullptr;  // pinned host staging for pipelined results
    size_t pinned_cap = 0;
    zl_ctx* aux2 = nullptr;  // second auxiliary context (Groth16: the witness map runs beside the witness-only MSMs)
    zl_ctx* stream_lane_ctx = nullptr;  // the fork zl_groth16_prove_circuits keeps for its second host thread (zl_ctx_drop_lanes / zl_ctx_destroy release it)
    zl_ctx* aux = nullptr;  // auxiliary stream + scratch set (Groth16: the G2 MSM overlaps the G1 MSMs)
    void* fb_table[4] = {nullptr, nullptr, nullptr, nullptr};  // fixed-base window tables of the generators (per group config, built on first use)
    zl_worker* workers[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // persistent host threads of this ctx: 0 witness-map issue, 1 G2 MSM (Groth16); 2..5 one per lane stream (side-by-side MSM batches)
    void* g16_h = nullptr;  // quotient polynomial of the last zl_groth16_prove (inside scratch slot 8; reset when the next proof starts)
    size_t g16_h_n = 0;
    int ntt_fit_beside = 0;  // this ctx's transforms run beside a kernel that leaves 96 registers per SIMD (a proof's witness map: ctx->aux2): use the capped passes (zl_ntt.hip)
    void* g16_z = nullptr;  // the canonical assignment of the last witness-map-only run (sharded proofs: the other ranks copy their slices from here)
    // MEASUREMENT ONLY (zl_test_acc_clock): when non-null, large G1 accumulations run as k_msm_accumulate_clk and leave four clock reads per wave here
    void* acc_clk = nullptr;
    size_t acc_clk_cap = 0, acc_clk_waves = 0;
};

// handle lookup: a ctx's own objects first, then its parent's (forks).  The maps of a ctx with live forks are read-only (uploads and frees are refused), so
// concurrent finds from several lanes are plain concurrent reads of a std::map.
inline const zl_bases* zl_find_bases(const zl_ctx* ctx, uint64_t handle) {
    for (const zl_ctx* c = ctx; c; c = c->parent) {
        std::shared_lock<std::shared_mutex> lk(c->maps_mu);
        auto it = c->bases.find(handle);
        if (it != c->bases.end()) return &it->second;
    }
    return nullptr;
}
inline const zl_r1cs_dev* zl_find_r1cs(const zl_ctx* ctx, uint64_t handle) {
    for (const zl_ctx* c = ctx; c; c = c->parent) {
        std::shared_lock<std::shared_mutex> lk(c->maps_mu);
        auto it = c->r1cs.find(handle);
        if (it != c->r1cs.end()) return &it->second;
    }
    return nullptr;
}
// the lazily built per-handle caches (zl_bases::d_endo, first_xy) are filled under this lock: two lanes may meet on the first use of a shared handle
inline std::mutex& zl_bases_cache_mutex() {
    static std::mutex m;
    return m;
}

// the auxiliary contexts of a ctx (Groth16: aux = G2 MSM, aux2 = witness map) and EVERY stream the ctx will ever use, created together in one fixed order (zl_capi.hip)
int zl_ctx_aux_init(zl_ctx* ctx);
int zl_ctx_streams_init(zl_ctx* ctx);

// The fork zl_groth16_prove_circuits keeps for its second host thread is the LIBRARY's, not the caller's: it must not pin the parent's handles once the call
// that used it has returned (zl_bases_free / zl_r1cs_free / zl_bases_precompute on the root ctx returned ZL_EINVAL after one prove_many(),
// and the callers that ignored the code leaked the whole device-resident key).  Every entry that refuses to run beside a live fork releases that idle lane first;
// forks the CALLER made (zl_ctx_fork) still pin the handles, as documented.
inline void zl_ctx_release_idle_lane(zl_ctx* ctx) {
    if (ctx && ctx->stream_lane_ctx && !ctx->pipeline_busy.load()) (void)zl_ctx_drop_lanes(ctx);
}

// developer tuning knob / deployment limit read from the environment (unset = the default)
inline int zl_tune(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
// the first `count` events of the ctx's pool of one kind (grown on demand; owned by the ctx: zl_ctx_destroy).  The pointer is valid until the next call for
// the SAME kind that has to grow the pool: callers take it once per pipeline call, under ctx->pipeline_busy
inline int zl_ctx_events(zl_ctx* ctx, int timing, size_t count, hipEvent_t** out) {
    std::vector<hipEvent_t>& pool = ctx->ev_pool[timing ? 1 : 0];
    while (pool.size() < count) {
        hipEvent_t e = nullptr;
        const hipError_t he = timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (he != hipSuccess) { ctx->last_hip = (int)he; return ZL_EHIP; }
        pool.push_back(e);
    }
    *out = pool.data();
    return ZL_OK;
}
inline zl_worker& zl_ctx_worker(zl_ctx* ctx, int k) {
    if (!ctx->workers[k]) ctx->workers[k] = new zl_worker();
    return *ctx->workers[k];
}
// grow-only device scratch slot
inline int zl_scratch_get(zl_ctx* ctx, int slot, size_t bytes, void** out) {
    zl_scratch& s = ctx->scratch[slot];
    if (s.cap < bytes) {
        if (s.p) {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
            (void)hipFree(s.p);
            s.p = nullptr;
            s.cap = 0;
        }
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&s.p, want);
        if (e != hipSuccess) { ctx->last_hip = (int)e; s.p = nullptr; return ZL_ENOMEM; }
        s.cap = want;
    }
    *out = s.p;
    return ZL_OK;
}

// entry points implemented per translation unit; the MSM file is compiled once per group (suffix = group config)
#define ZL_DECL_GROUP(G)                                                                                                   \
    int zl_msm_run_##G(zl_ctx* ctx, const zl_bases& b, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial); \
    int zl_msm_run_batch_##G(zl_ctx* ctx, const zl_bases& b, size_t first, const void* const* d_scalars, size_t n, size_t count, uint64_t* out_partials); \
    int zl_msm_run_jobs_##G(zl_ctx* ctx, const zl_bases* const* bases, const size_t* first, const void* const* d_scalars, const size_t* n, const hipEvent_t* wait, size_t count, uint64_t* out_partials, const std::atomic<int>* recorded, const std::function<void(size_t)>* on_done); \
    int zl_partial_to_affine_##G(const uint64_t* partial, uint64_t* out_xy, uint8_t* out_inf);                              \
    int zl_partials_fold_##G(const uint64_t* partials, size_t count, uint64_t* out_partial);                                \
    int zl_partial_from_affine_##G(const uint64_t* xy, uint64_t* out_partial);                                \
    int zl_bases_upload_##G(zl_ctx* ctx, const void* xy, size_t n, size_t stride, long inf_off, unsigned flags, zl_bases* out); \
    int zl_bases_generate_##G(zl_ctx* ctx, const uint64_t* k, size_t n, zl_bases* out);                                     \
    int zl_bases_download_##G(zl_ctx* ctx, const zl_bases& b, size_t first, size_t count, uint64_t* out_xy);                \
    int zl_bases_precompute_##G(zl_ctx* ctx, zl_bases& b, int c);                                                           \
    int zl_bases_concat_##G(zl_ctx* ctx, const zl_bases* const* parts, const size_t* first, const size_t* n, size_t count, zl_bases* out);
ZL_DECL_GROUP(BlsG1)
ZL_DECL_GROUP(BnG1)
ZL_DECL_GROUP(BlsG2)
ZL_DECL_GROUP(BnG2)
// dispatch on (curve, group)
#define ZL_DISPATCH(curve, group, fn, ...)                                                   \
    (((curve) == ZL_BLS12_381 && (group) == ZL_G1)   ? fn##_BlsG1(__VA_ARGS__)               \
     : ((curve) == ZL_BN254 && (group) == ZL_G1)     ? fn##_BnG1(__VA_ARGS__)                \
     : ((curve) == ZL_BLS12_381 && (group) == ZL_G2) ? fn##_BlsG2(__VA_ARGS__)               \
     : ((curve) == ZL_BN254 && (group) == ZL_G2)     ? fn##_BnG2(__VA_ARGS__)                \
                                                     : (int)ZL_EINVAL)
int zl_ntt_run(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned flags);
int zl_ntt_run_batch(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned flags, unsigned count, size_t stride_bytes);  // `count` equal transforms, one launch per pass
int zl_ntt_cross_run(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned log_g, unsigned rank, unsigned flags);
void zl_ntt_free(zl_ctx* ctx);
