// zl_capi.hip -- the extern "C" surface declared in include/zl_backend.h (context, bases, MSM, NTT wrappers).
#include <stdio.h>
#include <string.h>
#include <new>
#include "zl_ctx.h"

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
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&ctx->ev[i]);
    if (e != hipSuccess) {
        for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return ZL_EHIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return ZL_OK;
}

void zl_ctx_destroy(zl_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->bases) {
        if (kv.second.d_pts) (void)hipFree(kv.second.d_pts);
        if (kv.second.d_table) (void)hipFree(kv.second.d_table);
        if (kv.second.d_inf) (void)hipFree(kv.second.d_inf);
    }
    for (auto& kv : ctx->r1cs) if (kv.second.d_base) (void)hipFree(kv.second.d_base);
    for (auto& s : ctx->scratch) if (s.p) (void)hipFree(s.p);
    for (auto& t : ctx->fb_table) if (t) (void)hipFree(t);
    if (ctx->stream_sort) (void)hipStreamDestroy(ctx->stream_sort);
    if (ctx->stream_tail) (void)hipStreamDestroy(ctx->stream_tail);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    zl_ntt_free(ctx);
    for (zl_ctx** ax : {&ctx->aux, &ctx->aux2}) {
        zl_ctx* a = *ax;
        if (!a) continue;
        (void)hipStreamSynchronize(a->stream);
        for (auto& sc : a->scratch) if (sc.p) (void)hipFree(sc.p);
        if (a->stream_sort) (void)hipStreamDestroy(a->stream_sort);
        if (a->stream_tail) (void)hipStreamDestroy(a->stream_tail);
        if (a->pinned) (void)hipHostFree(a->pinned);
        zl_ntt_free(a);
        for (auto& ev : a->ev) if (ev) (void)hipEventDestroy(ev);
        if (a->own_stream) (void)hipStreamDestroy(a->own_stream);
        delete a;
        *ax = nullptr;
    }
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
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
    const uint64_t h = ctx->next_handle++;
    ctx->bases[h] = b;
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
    const uint64_t h = ctx->next_handle++;
    ctx->bases[h] = b;
    *handle_out = h;
    return ZL_OK;
}
int zl_bases_download(zl_ctx* ctx, uint64_t handle, size_t first, size_t count, uint64_t* out_xy) {
    if (!ctx || (!out_xy && count)) return ZL_EINVAL;
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    if (first > it->second.n || count > it->second.n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(it->second.curve, it->second.group, zl_bases_download, ctx, it->second, first, count, out_xy);
}
int zl_bases_precompute(zl_ctx* ctx, uint64_t handle, int c) {
    if (!ctx || c < 0) return ZL_EINVAL;
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(it->second.curve, it->second.group, zl_bases_precompute, ctx, it->second, c);
}
int zl_bases_free(zl_ctx* ctx, uint64_t handle) {
    if (!ctx) return ZL_EINVAL;
    auto it = ctx->bases.find(handle);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (it->second.d_pts) (void)hipFree(it->second.d_pts);
    if (it->second.d_table) (void)hipFree(it->second.d_table);
    if (it->second.d_inf) (void)hipFree(it->second.d_inf);
    ctx->bases.erase(it);
    return ZL_OK;
}

int zl_msm_batch_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* const* d_scalars, size_t n, size_t count, uint64_t* out_partials) {
    if (!ctx || (count && (!out_partials || !d_scalars))) return ZL_EINVAL;
    for (size_t i = 0; i < count; i++) if (!d_scalars[i] && n) return ZL_EINVAL;
    auto it = ctx->bases.find(bases);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    if (first > it->second.n || n > it->second.n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(it->second.curve, it->second.group, zl_msm_run_batch, ctx, it->second, first, d_scalars, n, count, out_partials);
}
int zl_msm_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial) {
    if (!ctx || !out_partial || (!d_scalars && n)) return ZL_EINVAL;
    auto it = ctx->bases.find(bases);
    if (it == ctx->bases.end()) return ZL_EHANDLE;
    if (first > it->second.n || n > it->second.n - first) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return ZL_DISPATCH(it->second.curve, it->second.group, zl_msm_run, ctx, it->second, first, d_scalars, n, out_partial);
}
int zl_msm_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!out_xy) return ZL_EINVAL;
    uint64_t partial[ZL_PARTIAL_WORDS];
    int rc = zl_msm_partial_dev(ctx, bases, first, d_scalars, n, partial);
    if (rc) return rc;
    const zl_bases& b = ctx->bases[bases];
    return ZL_DISPATCH(b.curve, b.group, zl_partial_to_affine, partial, out_xy, out_inf);
}
int zl_msm(zl_ctx* ctx, uint64_t bases, size_t first, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
    if (!ctx || !out_xy || (!scalars && n)) return ZL_EINVAL;
    if (ctx->bases.find(bases) == ctx->bases.end()) return ZL_EHANDLE;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    void* d_sc = nullptr;
    if (n) {
        int rc = zl_scratch_get(ctx, 7, n * 32, &d_sc);
        if (rc) return rc;
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
