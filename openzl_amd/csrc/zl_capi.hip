// zl_capi.hip -- the extern "C" surface declared in include/zl_backend.h (context, bases, MSM, NTT wrappers).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <thread>
#include <vector>
#include "zl_ctx.h"
#include "zl_pool.h"

zl_pool& zl_pool_get() {
    static zl_pool pool(std::min(7u, std::max(1u, std::thread::hardware_concurrency()) - 1u));  // + the calling thread
    return pool;
}

// aux (G2 MSM): default priority, like the G1 accumulation stream (the short sort / tail kernels of both run on highest-priority streams, so nothing waits behind
// the long G2 accumulate; lowest priority measured 1 ms slower); aux2 (witness map): highest, h gates the last MSM
int zl_ctx_aux_init(zl_ctx* ctx) {
    for (zl_ctx** ax : {&ctx->aux, &ctx->aux2}) {
        if (*ax) continue;
        zl_ctx* a = new (std::nothrow) zl_ctx();
        if (!a) return ZL_ENOMEM;
        a->device = ctx->device;
        a->cu_count = ctx->cu_count;
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        hipError_t e = hipStreamCreateWithPriority(&a->own_stream, hipStreamNonBlocking, ax == &ctx->aux ? 0 : prio_hi);
        for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&a->ev[i]);
        a->stream = a->own_stream;
        *ax = a;  // owned by ctx from here on (zl_ctx_destroy)
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
    }
    return ZL_OK;
}
// Every stream of the ctx at once, in one fixed order.  The runtime multiplexes the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4)
// by the order in which they come into being, and two kernel chains on one hardware queue run in turn (tools/queue_chains.hip) -- created lazily, the lanes of a
// small proof landed on different queues depending on which legs of a process had run before (bench.py after its MSM legs: 235 constraints 1.5-1.8 ms; a fresh
// process: 1.05-1.3).  With the whole population created here, a ctx behaves the same whatever it was used for first.
int zl_ctx_streams_init(zl_ctx* ctx) {
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // The order is the one round 5's sweep of five creation orders found best at all three proof sizes (profiles/r05_stream_order_ab.log): the G2 MSM's and the
    // witness map's streams, then the four lanes (default class), then sort / tails / copy (high class).  It depends on the HIP runtime's creation-order -> queue
    // mapping and is not tuned further; the sweep's knobs were removed in round 6.
    int rc = ZL_OK;
    if ((rc = zl_ctx_aux_init(ctx))) return rc;
    for (int k = 0; k < 4; k++) if (!ctx->stream_lane[k]) ZL_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_lane[k], hipStreamNonBlocking));
    if (!ctx->stream_sort) ZL_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream_sort, hipStreamNonBlocking, prio_hi));
    for (auto& t : ctx->stream_tail) if (!t) ZL_HIP(ctx, hipStreamCreateWithPriority(&t, hipStreamNonBlocking, prio_hi));
    if (!ctx->stream_copy) ZL_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream_copy, hipStreamNonBlocking, prio_hi));
    return ZL_OK;
}

extern "C" {

const char* zl_strerror(int code) {
    switch (code) {
    case ZL_OK: return "ok";
    case ZL_EINVAL: return "invalid argument";
    case ZL_ENOMEM: return "out of memory";
    case ZL_EHIP: return "HIP runtime error";
    case ZL_ENODEV: return "no usable gfx950 device";
    case ZL_EHANDLE: return "unknown bases handle";
    case ZL_ENOTCURVE: return "base point not on curve";
    default: return "unknown error";
    }
}

int zl_ctx_create(zl_ctx** out, int device_id) {
    if (!out) return ZL_EINVAL;
    *out = nullptr;
    // (The library runs up to ~12 streams at once -- pipeline phases, side-by-side lanes, G2 MSM, witness map, tails -- which the HIP runtime
    // multiplexes onto GPU_MAX_HW_QUEUES hardware queues, default 4.  Measured with 8: a 235-constraint proof 1.26 -> 1.11 ms (median), the
    // 958 465-constraint proof 19.5 -> 20.25 ms (the G2 accumulation, one 416-register wave per SIMD, then runs beside the G1 accumulation instead
    // of between two of them and both lose occupancy).  The default stays; a deployment of small circuits can set the variable itself.
    // Round 5: with every stream of the ctx created here in one fixed order (zl_ctx_streams_init) the default of 4 beats 8 at every proof size -- 18.3 / 2.25 / 1.10 ms
    // against 19.2 / 2.65 / 1.19 ms for 958 465 / 14 977 / 235 constraints: nothing to set.)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return ZL_ENODEV;
    if (device_id < 0 || device_id >= count) return ZL_EINVAL;
    zl_ctx* ctx = new (std::nothrow) zl_ctx();
    if (!ctx) return ZL_ENOMEM;
    ctx->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_id);
    if (e == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // the kernels are built for gfx950 only
        delete ctx;
        return ZL_ENODEV;
    }
    if (e == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return ZL_EHIP;
    }
    ctx->stream = ctx->own_stream;
    const int rc_s = zl_ctx_streams_init(ctx);
    if (rc_s) { zl_ctx_destroy(ctx); return rc_s; }
    *out = ctx;
    return ZL_OK;
}

int zl_ctx_fork(zl_ctx* parent, zl_ctx** out) {
    if (!parent || !out) return ZL_EINVAL;
    *out = nullptr;
    if (parent->parent) return ZL_EINVAL;  // one level: fork the root
    zl_ctx* c = nullptr;
    const int rc = zl_ctx_create(&c, parent->device);
    if (rc) return rc;
    c->msm_c = parent->msm_c;
    {
        std::unique_lock<std::shared_mutex> lk(parent->maps_mu);  // (two threads may fork one parent at once)
        c->parent = parent;
        c->next_handle = ((uint64_t)(++parent->fork_seq) << 48) | 1;
        parent->fork_list.push_back(c);
        parent->forks.fetch_add(1);
    }
    *out = c;
    return ZL_OK;
}

int zl_ctx_drop_lanes(zl_ctx* ctx) {
    if (!ctx) return ZL_EINVAL;
    if (ctx->pipeline_busy.load()) return ZL_EINVAL;
    zl_ctx* l = ctx->stream_lane_ctx;
    if (l && l->pipeline_busy.load()) return ZL_EINVAL;  // (the lane itself may be inside a call)
    ctx->stream_lane_ctx = nullptr;
    if (l) zl_ctx_destroy(l);
    return ZL_OK;
}
void zl_ctx_destroy(zl_ctx* ctx) {
    if (!ctx) return;
    if (ctx->stream_lane_ctx) { zl_ctx* l = ctx->stream_lane_ctx; ctx->stream_lane_ctx = nullptr; zl_ctx_destroy(l); }
    if (ctx->parent) {
        std::unique_lock<std::shared_mutex> lk(ctx->parent->maps_mu);
        auto& fl = ctx->parent->fork_list;
        for (size_t i = 0; i < fl.size(); i++) if (fl[i] == ctx) { fl.erase(fl.begin() + (long)i); break; }
        ctx->parent->forks.fetch_sub(1);
    }
    for (zl_ctx* f : ctx->fork_list) f->parent = nullptr;  // destroyed before its forks (a caller error): orphan them rather than leave them a dangling pointer
    ctx->fork_list.clear();
    (void)hipSetDevice(ctx->device);
    for (zl_ctx* c : {ctx, ctx->aux, ctx->aux2}) {
        if (!c) continue;
        for (auto& w : c->workers) { delete w; w = nullptr; }  // (joins: a worker is idle between calls)
    }
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->bases) {
        if (kv.second.d_pts) (void)hipFree(kv.second.d_pts);
        if (kv.second.d_table) (void)hipFree(kv.second.d_table);
        if (kv.second.d_endo) (void)hipFree(kv.second.d_endo);
        if (kv.second.d_inf) (void)hipFree(kv.second.d_inf);
    }
    for (auto& kv : ctx->r1cs) if (kv.second.d_base) (void)hipFree(kv.second.d_base);
    for (auto& s : ctx->scratch) if (s.p) (void)hipFree(s.p);
    for (auto& t : ctx->fb_table) if (t) (void)hipFree(t);
    if (ctx->acc_clk) (void)hipFree(ctx->acc_clk);
    if (ctx->stream_sort) (void)hipStreamDestroy(ctx->stream_sort);
    for (auto& t : ctx->stream_tail) if (t) (void)hipStreamDestroy(t);
    for (auto& t : ctx->stream_lane) if (t) (void)hipStreamDestroy(t);
    if (ctx->stream_copy) (void)hipStreamDestroy(ctx->stream_copy);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    zl_ntt_free(ctx);
    for (zl_ctx** ax : {&ctx->aux, &ctx->aux2}) {
        zl_ctx* a = *ax;
        if (!a) continue;
        (void)hipStreamSynchronize(a->stream);
        for (auto& sc : a->scratch) if (sc.p) (void)hipFree(sc.p);
        if (a->stream_sort) (void)hipStreamDestroy(a->stream_sort);
        for (auto& t : a->stream_tail) if (t) (void)hipStreamDestroy(t);
        for (auto& t : a->stream_lane) if (t) (void)hipStreamDestroy(t);
        if (a->pinned) (void)hipHostFree(a->pinned);
        zl_ntt_free(a);
        for (auto& ev : a->ev) if (ev) (void)hipEventDestroy(ev);
        for (auto& pool : a->ev_pool) for (auto& ev : pool) (void)hipEventDestroy(ev);
        if (a->own_stream) (void)hipStreamDestroy(a->own_stream);
        delete a;
        *ax = nullptr;
    }
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& pool : ctx->ev_pool) for (auto& ev : pool) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int zl_ctx_set_stream(zl_ctx* ctx, void* hip_stream) {
    if (!ctx) return ZL_EINVAL;
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return ZL_OK;
}
int zl_ctx_sync(zl_ctx* ctx) {
    if (!ctx) return ZL_EINVAL;
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZL_OK;
}
int zl_ctx_set_msm_window(zl_ctx* ctx, int c) {
    if (!ctx || c < 0 || c > 22 || c == 1) return ZL_EINVAL;
    ctx->msm_c = c;
    return ZL_OK;
}
int zl_ctx_last_hip_error(const zl_ctx* ctx) { return ctx ? ctx->last_hip : 0; }
int zl_ctx_enable_timing(zl_ctx* ctx, int on) {
    if (!ctx) return ZL_EINVAL;
    ctx->timing_on = on ? 1 : 0;
    return ZL_OK;
}
int zl_last_timing(zl_ctx* ctx, zl_timing* out) {
    if (!ctx || !out) return ZL_EINVAL;
    *out = ctx->timing;
    return ZL_OK;
}
int zl_describe(zl_ctx* ctx, char* buf, size_t buflen) {
    if (!ctx || !buf || !buflen) return ZL_EINVAL;
    hipDeviceProp_t prop;
    ZL_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    int n = snprintf(buf, buflen, "libzl_backend gfx950 HIP backend; device %d: %s arch=%s CUs=%d LDS/block=%zu HBM=%.1f GiB", ctx->device, prop.name,
                     prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock, (double)prop.totalGlobalMem / (1 << 30));
    return n;
}

static bool valid_cg(zl_curve_t curve, zl_group_t group) {
    return (curve == ZL_BLS12_381 || curve == ZL_BN254) && (group == ZL_G1 || group == ZL_G2);
}

int zl_bases_upload(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const void* xy, size_t n, size_t stride_bytes, long inf_offset, unsigned flags,
                    uint64_t* handle_out) {
    if (!ctx || !handle_out || (!xy && n) || !valid_cg(curve, group)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    zl_bases b;
    b.curve = curve;
    b.group = group;
    int rc = ZL_DISPATCH(curve, group, zl_bases_upload, ctx, xy, n, stride_bytes, inf_offset, flags, &b);
    if (rc) return rc;
    uint64_t h;
    {
        std::unique_lock<std::shared_mutex> lk(ctx->maps_mu);
        h = ctx->next_handle++;
        ctx->bases[h] = b;
    }
    *handle_out = h;
    return ZL_OK;
}
int zl_bases_generate(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const uint64_t* k, size_t n, uint64_t* handle_out) {
    if (!ctx || !handle_out || (!k && n) || !valid_cg(curve, group)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    zl_bases b;
    b.curve = curve;
    b.group = group;
    int rc = ZL_DISPATCH(curve, group, zl_bases_generate, ctx, k, n, &b);
    if (rc) return rc;
    uint64_t h;
    {
        std::unique_lock<std::shared_mutex> lk(ctx->maps_mu);
        h = ctx->next_handle++;
        ctx->bases[h] = b;
    }
    *handle_out = h;
    return ZL_OK;
}
int zl_bases_download(zl_ctx* ctx, uint64_t handle, size_t first, size_t count, uint64_t* out_xy) {
    if (!ctx || (!out_xy && count)) return ZL_EINVAL;
    const zl_bases* bp = zl_find_bases(ctx, handle);
    if (!bp) return ZL_EHANDLE;
    if (first > bp->n || count > bp->n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(bp->curve, bp->group, zl_bases_download, ctx, *bp, first, count, out_xy);
}
int zl_bases_precompute(zl_ctx* ctx, uint64_t handle, int c) {
    if (!ctx || c < 0) return ZL_EINVAL;
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    zl_ctx_release_idle_lane(ctx);
    if (ctx->forks.load() > 0) return ZL_EINVAL;  // the table replaces what the lanes read: build it before forking
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(it->second.curve, it->second.group, zl_bases_precompute, ctx, it->second, c);
}
int zl_bases_free(zl_ctx* ctx, uint64_t handle) {
    if (!ctx) return ZL_EINVAL;
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    zl_ctx_release_idle_lane(ctx);
    if (ctx->forks.load() > 0) return ZL_EINVAL;  // a fork may be reading it: destroy the forks first
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (it->second.d_pts) (void)hipFree(it->second.d_pts);
    if (it->second.d_table) (void)hipFree(it->second.d_table);
    if (it->second.d_endo) (void)hipFree(it->second.d_endo);
    if (it->second.d_inf) (void)hipFree(it->second.d_inf);
    ctx->bases.erase(it);
    return ZL_OK;
}

int zl_msm_batch_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* const* d_scalars, size_t n, size_t count, uint64_t* out_partials) {
    if (!ctx || (count && (!out_partials || !d_scalars))) return ZL_EINVAL;
    for (size_t i = 0; i < count; i++) if (!d_scalars[i] && n) return ZL_EINVAL;
    const zl_bases* bp = zl_find_bases(ctx, bases);
    if (!bp) return ZL_EHANDLE;
    if (first > bp->n || n > bp->n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(bp->curve, bp->group, zl_msm_run_batch, ctx, *bp, first, d_scalars, n, count, out_partials);
}
int zl_msm_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial) {
    if (!ctx || !out_partial || (!d_scalars && n)) return ZL_EINVAL;
    const zl_bases* bp = zl_find_bases(ctx, bases);
    if (!bp) return ZL_EHANDLE;
    if (first > bp->n || n > bp->n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(bp->curve, bp->group, zl_msm_run, ctx, *bp, first, d_scalars, n, out_partial);
}
int zl_msm_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!out_xy) return ZL_EINVAL;
    uint64_t partial[ZL_PARTIAL_WORDS];
    int rc = zl_msm_partial_dev(ctx, bases, first, d_scalars, n, partial);
    if (rc) return rc;
    const zl_bases& b = *zl_find_bases(ctx, bases);  // (found by zl_msm_partial_dev above)
    return ZL_DISPATCH(b.curve, b.group, zl_partial_to_affine, partial, out_xy, out_inf);
}
// Host scalars (what VariableBaseMSM::multi_scalar_mul is handed: the witness is new for every proof).  One copy followed by one MSM leaves
// the whole transfer (32 B / point: 12.6 ms for 2^24 from pageable memory) in front of the first kernel.  Large inputs are therefore cut into
// a few growing shards of points: MSM(all) = sum of the shards' MSMs, a helper thread copies shard j + 1 on its own stream while the pipeline of
// zl_msm_batch_partial_dev works on the shards before it (each job waits for its copy's event), and only the first, small copy stays exposed.
// Shard plan (round 4, profiles/r04_host_shards.log): 2^20, then four times the previous shard, the rest as the last one -- 2^20, 2^22, 11.5 M at
// 2^24.  The copy of a shard (1.3 M scalars / ms) must hide under the MSM of the one before it (0.3 - 0.43 M points / ms), which allows a growth of
// 3 - 4x; every extra shard costs its own bucket sets and narrower windows.  Round 3's doubling plan (2^20, 2^20, 2^21, 2^22, 2^23) measured 44.5 ms
// against 41.8 for this one (38.4 with the scalars already on the device); 2^21 first 43.0; 2^20, 2^21, 2^22, rest 42.9; 2^20, 2^22, 2^23, rest 42.7.
static int msm_host_chunked(zl_ctx* ctx, const zl_bases& b, size_t first, const uint64_t* scalars, size_t n, void* d_sc, uint64_t* out_xy, uint8_t* out_inf) {
    std::vector<size_t> off, len;
    size_t done = 0, step = (size_t)1 << 20;
    const char* plan = getenv("ZL_TUNE_HOST_SHARDS");  // developer sweep: log2 sizes of the leading shards, e.g. "20,22" (the rest is one shard)
    while (done < n) {
        size_t l = std::min(step, n - done);
        if (plan) {
            l = *plan ? std::min<size_t>((size_t)1 << atoi(plan), n - done) : n - done;
            while (*plan && *plan != ',') plan++;
            if (*plan == ',') plan++;
        }
        if (n - done - l < step / 2) l = n - done;  // no small last shard
        off.push_back(done);
        len.push_back(l);
        done += l;
        step <<= 2;
    }
    const size_t K = off.size();
    if (!ctx->stream_copy) {
        // In the priority class of the sort / tail streams, NOT the default one: the runtime spreads the default-priority streams of a process over a few
        // hardware queues in creation order, and one more of them changes which streams of a later Groth16 proof share a queue (its G2 MSM and one of the four
        // G1 lanes: +1 ms on the 958 465-constraint proof after a single zl_msm call with host scalars, profiles/r04_g16_hwq_ab.log).  The stream only carries
        // H2D copies; priority means nothing to them.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        ZL_HIP(ctx, hipStreamCreateWithPriority(&ctx->stream_copy, hipStreamNonBlocking, prio_hi));
    }
    std::vector<hipEvent_t> ev(K, nullptr);
    for (size_t j = 0; j < K; j++) {
        const hipError_t e = hipEventCreateWithFlags(&ev[j], hipEventDisableTiming);
        if (e != hipSuccess) { for (auto x : ev) if (x) (void)hipEventDestroy(x); ctx->last_hip = (int)e; return ZL_EHIP; }
    }
    std::atomic<int> recorded{0};
    const int dev = ctx->device;
    hipStream_t sc = ctx->stream_copy;
    std::thread copier([&]() {
        hipError_t e = hipSetDevice(dev);
        for (size_t j = 0; j < K && e == hipSuccess; j++) {
            // from pageable memory this call stages through the runtime's pinned buffers and returns when the last piece is queued
            e = hipMemcpyAsync(reinterpret_cast<unsigned char*>(d_sc) + off[j] * 32, reinterpret_cast<const unsigned char*>(scalars) + off[j] * 32, len[j] * 32,
                               hipMemcpyHostToDevice, sc);
            if (e == hipSuccess) e = hipEventRecord(ev[j], sc);
            if (e == hipSuccess) recorded.store((int)j + 1, std::memory_order_release);
        }
        if (e != hipSuccess) recorded.store(-1, std::memory_order_release);
    });
    std::vector<const zl_bases*> jb(K, &b);
    std::vector<size_t> jf(K), jn(K);
    std::vector<const void*> js(K);
    for (size_t j = 0; j < K; j++) { jf[j] = first + off[j]; jn[j] = len[j]; js[j] = reinterpret_cast<unsigned char*>(d_sc) + off[j] * 32; }
    std::vector<uint64_t> parts(K * ZL_PARTIAL_WORDS);
    // (Round 5 built the alternative the earlier rounds had only costed -- the shards as jobs over ONE carried bucket set with the windows of the whole MSM -- and
    // measured it slower for every shard plan tried, 42.6-42.9 ms against 40.4-40.9 for these independent jobs: a 2^20-point shard spread over the 3.67 M buckets
    // of c = 19 has 4 entries per bucket.  profiles/r05_host_carry_ab.log; the code was removed in round 6.)
    const int rc = ZL_DISPATCH(b.curve, b.group, zl_msm_run_jobs, ctx, jb.data(), jf.data(), js.data(), jn.data(), ev.data(), K, parts.data(), &recorded, (const std::function<void(size_t)>*)nullptr);
    copier.join();
    (void)hipStreamSynchronize(sc);
    for (auto x : ev) (void)hipEventDestroy(x);
    if (rc) return rc;
    return zl_partials_sum((zl_curve_t)b.curve, (zl_group_t)b.group, parts.data(), K, out_xy, out_inf);
}

int zl_msm(zl_ctx* ctx, uint64_t bases, size_t first, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx || !out_xy || (!scalars && n)) return ZL_EINVAL;
    const zl_bases* bp = zl_find_bases(ctx, bases);
    if (!bp) return ZL_EHANDLE;
    if (first > bp->n || n > bp->n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    void* d_sc = nullptr;
    if (n) {
        int rc = zl_scratch_get(ctx, 7, n * 32, &d_sc);
        if (rc) return rc;
        // the window table of a precomputed handle is built for full-size MSMs over it; shards would each pay its merged bucket set
        // (ZL_TUNE_HOST_CHUNK_MIN_LOG: developer / test knob -- the shard pipeline at sizes the oracle can check, with ZL_TUNE_HOST_SHARDS naming the shards)
        if (n >= ((size_t)1 << zl_tune("ZL_TUNE_HOST_CHUNK_MIN_LOG", 22)) && bp->precomp_c == 0 && !getenv("ZL_NO_HOST_CHUNKS")) return msm_host_chunked(ctx, *bp, first, scalars, n, d_sc, out_xy, out_inf);
        ZL_HIP(ctx, hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    return zl_msm_dev(ctx, bases, first, d_sc, n, out_xy, out_inf);
}

int zl_partials_sum(zl_curve_t curve, zl_group_t group, const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
    if ((!partials && count) || !out_xy || !valid_cg(curve, group)) return ZL_EINVAL;
    uint64_t acc[ZL_PARTIAL_WORDS];
    int rc = ZL_DISPATCH(curve, group, zl_partials_fold, partials, count, acc);
    if (rc) return rc;
    return ZL_DISPATCH(curve, group, zl_partial_to_affine, acc, out_xy, out_inf);
}

int zl_partial_from_affine(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint64_t* out_partial) {
    if (!xy || !out_partial || !valid_cg(curve, group)) return ZL_EINVAL;
    return ZL_DISPATCH(curve, group, zl_partial_from_affine, xy, out_partial);
}

int zl_ntt_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned flags) {
    if (!ctx || !d_data) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    if (flags & ~(ZL_MONT | ZL_COSET | ZL_INVERSE | ZL_MONT_IN | ZL_MONT_OUT)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return zl_ntt_run(ctx, curve, d_data, log_n, flags);
}
int zl_ntt_batch_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned flags, unsigned count, size_t stride_elems) {
    if (!ctx || !d_data) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    if (flags & ~(ZL_MONT | ZL_COSET | ZL_INVERSE | ZL_MONT_IN | ZL_MONT_OUT)) return ZL_EINVAL;
    if (log_n > 30 || count > 65535 || (count > 1 && stride_elems < ((size_t)1 << log_n))) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return zl_ntt_run_batch(ctx, curve, d_data, log_n, flags, count, stride_elems * 32);
}
int zl_ntt_cross_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned log_g, unsigned rank, unsigned flags) {
    if (!ctx || !d_data) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    if (flags & ~(ZL_MONT | ZL_COSET | ZL_INVERSE | ZL_MONT_IN | ZL_MONT_OUT)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return zl_ntt_cross_run(ctx, curve, d_data, log_n, log_g, rank, flags);
}
int zl_ntt(zl_ctx* ctx, zl_curve_t curve, uint64_t* data, unsigned log_n, unsigned flags) {
    if (!ctx || !data || log_n > 32) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    if (log_n > (curve == ZL_BLS12_381 ? 32u : 28u) || log_n > 30) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = ((size_t)1 << log_n) * 32;
    void* d;
    int rc = zl_scratch_get(ctx, 7, bytes, &d);
    if (rc) return rc;
    ZL_HIP(ctx, hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = zl_ntt_dev(ctx, curve, d, log_n, flags);
    if (rc) return rc;
    ZL_HIP(ctx, hipMemcpyAsync(data, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZL_OK;
}

}  // extern "C"
