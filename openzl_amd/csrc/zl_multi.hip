// zl_multi.hip -- multi-GPU entry points of the C ABI (include/zl_backend.h, "multi-GPU" section): one process drives G devices, each
// with its own zl_ctx (stream, scratch, bases handles); RCCL (ncclCommInitAll over xGMI) carries the exchange steps.
//
// Replaces nothing in the reference (arkworks has no distributed MSM / FFT); it is the sharding SURVEY.md §8e lays out for the hot path
// behind Groth16::prove (/root/reference/plugins/arkworks/src/groth16.rs:445-457):
//   MSM  shard (bases, scalars) by contiguous index range -> complete local Pippenger per device -> ncclAllGather of the
//        ZL_PARTIAL_WORDS-u64 partial sums (EC addition is not an RCCL reduction op: gather-then-add IS the reduce) -> fold
//   NTT  four-step factorisation with ONE all-to-all (grouped ncclSend / ncclRecv) between zl_ntt_cross_dev and the local zl_ntt_dev
// RCCL is loaded with dlopen on first use, so single-device users never touch it.  Device ids may repeat ("virtual ranks" sharing one
// GPU): RCCL refuses duplicate devices, so the exchange then uses device-to-device copies on the ranks' streams -- the arithmetic and
// the data movement pattern are identical, which is how the entry points are tested on a 1-GPU box.  openzl_amd/sharded.py keeps the
// one-process-per-GPU variant over torch.distributed for callers that already live in that world (bench.py --gpus N).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "zl_ctx.h"

// The handful of RCCL declarations this unit needs, stated locally (ABI of rccl.h, ROCm 7: ncclResult_t / ncclDataType_t are plain int enums,
// ncclComm_t an opaque pointer, ncclSuccess = 0, ncclUint64 = 5): the library is loaded with dlopen, so neither its header nor the
// library itself has to exist on a box that only ever uses one device -- zl_ctx_create_multi then returns ZL_ENODEV for distinct devices.
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef struct ncclComm* ncclComm_t;
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclUint64 = 5;

struct zl_rccl_api {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
#define ZL_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)); if (!field) { dlclose(lib); lib = nullptr; return false; }
        ZL_SYM(CommInitAll, "ncclCommInitAll")
        ZL_SYM(CommDestroy, "ncclCommDestroy")
        ZL_SYM(AllGather, "ncclAllGather")
        ZL_SYM(Send, "ncclSend")
        ZL_SYM(Recv, "ncclRecv")
        ZL_SYM(GroupStart, "ncclGroupStart")
        ZL_SYM(GroupEnd, "ncclGroupEnd")
#undef ZL_SYM
        return true;
    }
};

struct zl_mctx {
    int n = 0;
    std::vector<zl_ctx*> ctx;
    std::vector<int> dev;
    bool virt = true;  // ranks share devices (or n == 1 without RCCL): exchanges are device-to-device copies
    zl_rccl_api rccl;
    std::vector<ncclComm_t> comms;
    std::vector<void*> xbuf;      // per rank: exchange / gather buffer (grow-only)
    std::vector<size_t> xcap;
    std::vector<hipEvent_t> ev;   // per rank: "my leg is done" marker for the virtual exchange
    int last_rccl = 0;
};

static int mctx_buf(zl_mctx* m, int g, size_t bytes, void** out) {
    if (m->xcap[g] < bytes) {
        if (hipSetDevice(m->dev[g]) != hipSuccess) return ZL_EHIP;
        if (m->xbuf[g]) { (void)hipStreamSynchronize(m->ctx[g]->stream); (void)hipFree(m->xbuf[g]); m->xbuf[g] = nullptr; m->xcap[g] = 0; }
        if (hipMalloc(&m->xbuf[g], bytes) != hipSuccess) return ZL_ENOMEM;
        m->xcap[g] = bytes;
    }
    *out = m->xbuf[g];
    return ZL_OK;
}

// run fn(rank) on one host thread per rank (every rank drives its own device / stream), first error wins
template <class Fn>
static int per_rank(zl_mctx* m, Fn fn) {
    std::vector<int> rc(m->n, ZL_OK);
    if (m->n == 1) return fn(0);
    std::vector<std::thread> th;
    for (int g = 0; g < m->n; g++) th.emplace_back([&, g]() { rc[g] = (hipSetDevice(m->dev[g]) == hipSuccess) ? fn(g) : (int)ZL_EHIP; });
    for (auto& t : th) t.join();
    for (int g = 0; g < m->n; g++) if (rc[g]) return rc[g];
    return ZL_OK;
}

// after a failure with work already enqueued: no rank's stream may still be touching stack-owned host buffers or another rank's memory
static void drain_all(zl_mctx* m) {
    for (int g = 0; g < m->n; g++) {
        if (!m->ctx[g]) continue;
        if (hipSetDevice(m->dev[g]) == hipSuccess) (void)hipStreamSynchronize(m->ctx[g]->stream);
    }
    (void)hipGetLastError();
}
#define ZL_MHIP(m, ctx, call)                                          \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) {                                        \
            (ctx)->last_hip = (int)e_;                                 \
            drain_all(m);                                              \
            return e_ == hipErrorOutOfMemory ? ZL_ENOMEM : ZL_EHIP;    \
        }                                                              \
    } while (0)

extern "C" {

int zl_ctx_create_multi(zl_mctx** out, const int* device_ids, int n_devices) {
    if (!out || !device_ids || n_devices < 1 || n_devices > 16) return ZL_EINVAL;
    *out = nullptr;
    zl_mctx* m = new (std::nothrow) zl_mctx();
    if (!m) return ZL_ENOMEM;
    m->n = n_devices;
    m->dev.assign(device_ids, device_ids + n_devices);
    m->ctx.assign(n_devices, nullptr);
    m->xbuf.assign(n_devices, nullptr);
    m->xcap.assign(n_devices, 0);
    m->ev.assign(n_devices, nullptr);
    int rc = ZL_OK;
    for (int g = 0; g < n_devices && !rc; g++) {
        rc = zl_ctx_create(&m->ctx[g], device_ids[g]);
        if (!rc && hipEventCreateWithFlags(&m->ev[g], hipEventDisableTiming) != hipSuccess) rc = ZL_EHIP;
    }
    bool distinct = true;
    for (int a = 0; a < n_devices; a++)
        for (int b = a + 1; b < n_devices; b++) distinct = distinct && device_ids[a] != device_ids[b];
    // ZL_FORCE_RCCL: also open a (one-rank) communicator for a single device, so that the RCCL branch -- dlopen, ncclCommInitAll, grouped
    // ncclAllGather -- can at least be smoke-tested on a 1-GPU box
    if (!rc && distinct && (n_devices > 1 || getenv("ZL_FORCE_RCCL"))) {
        // real multi-GPU: one communicator per device, created together (ncclCommInitAll = the single-process form of ncclCommInitRank)
        if (!m->rccl.load()) rc = ZL_ENODEV;
        if (!rc) {
            m->comms.assign(n_devices, nullptr);
            const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), n_devices, device_ids);
            if (r != ncclSuccess) { m->last_rccl = (int)r; m->comms.clear(); rc = ZL_EHIP; }
            else m->virt = false;
        }
    }
    if (rc) { zl_mctx_destroy(m); return rc; }
    *out = m;
    return ZL_OK;
}

void zl_mctx_destroy(zl_mctx* m) {
    if (!m) return;
    for (size_t g = 0; g < m->comms.size(); g++)
        if (m->comms[g]) (void)m->rccl.CommDestroy(m->comms[g]);
    for (int g = 0; g < m->n; g++) {
        if (!m->ctx[g]) continue;  // creation stopped before this rank (e.g. a bad device id): nothing of it exists
        (void)hipSetDevice(m->dev[g]);
        (void)hipStreamSynchronize(m->ctx[g]->stream);
        if (m->xbuf[g]) (void)hipFree(m->xbuf[g]);
        if (m->ev[g]) (void)hipEventDestroy(m->ev[g]);
        zl_ctx_destroy(m->ctx[g]);
    }
    if (m->rccl.lib) dlclose(m->rccl.lib);
    (void)hipGetLastError();  // a failed creation must not leave a stale error for the next launch check of another ctx
    delete m;
}

int zl_mctx_size(const zl_mctx* m) { return m ? m->n : 0; }
zl_ctx* zl_mctx_ctx(zl_mctx* m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
int zl_mctx_uses_rccl(const zl_mctx* m) { return (m && !m->virt) ? 1 : 0; }

int zl_msm_sharded(zl_mctx* m, const uint64_t* bases, const size_t* first, const void* const* d_scalars, const size_t* n, uint64_t* out_xy,
                   uint8_t* out_inf) {
    if (!m || !bases || !d_scalars || !n || !out_xy) return ZL_EINVAL;
    const int G = m->n;
    int curve = 0, group = 0;
    for (int g = 0; g < G; g++) {
        auto it = m->ctx[g]->bases.find(bases[g]);
        if (it == m->ctx[g]->bases.end()) return ZL_EHANDLE;
        if (g == 0) { curve = it->second.curve; group = it->second.group; }
        else if (curve != it->second.curve || group != it->second.group) return ZL_EHANDLE;
    }
    // 1. complete local Pippenger on every device, concurrently -> one un-normalised partial sum per rank (host memory)
    std::vector<uint64_t> parts((size_t)G * ZL_PARTIAL_WORDS);
    int rc = per_rank(m, [&](int g) -> int {
        return zl_msm_partial_dev(m->ctx[g], bases[g], first ? first[g] : 0, d_scalars[g], n[g], &parts[(size_t)g * ZL_PARTIAL_WORDS]);
    });
    if (rc) return rc;
    // 2. all-gather of the partials: every rank ends up with all G of them (what a one-process-per-GPU deployment needs; here rank 0's
    //    copy is folded).  RCCL over xGMI on distinct devices, device-to-device copies for virtual ranks.
    const size_t pbytes = (size_t)ZL_PARTIAL_WORDS * 8;
    std::vector<void*> buf(G);
    for (int g = 0; g < G; g++)
        if ((rc = mctx_buf(m, g, (size_t)(G + 1) * pbytes, &buf[g]))) return rc;  // [0, G): gathered, [G]: this rank's contribution
    std::vector<uint64_t> gathered((size_t)G * ZL_PARTIAL_WORDS);  // declared before anything is enqueued: every failure below drains first
    rc = per_rank(m, [&](int g) -> int {
        unsigned char* b = reinterpret_cast<unsigned char*>(buf[g]);
        ZL_HIP(m->ctx[g], hipMemcpyAsync(b + (size_t)G * pbytes, &parts[(size_t)g * ZL_PARTIAL_WORDS], pbytes, hipMemcpyHostToDevice, m->ctx[g]->stream));
        if (m->virt) ZL_HIP(m->ctx[g], hipEventRecord(m->ev[g], m->ctx[g]->stream));
        return (int)ZL_OK;
    });
    if (rc) { drain_all(m); return rc; }
    if (!m->virt) {
        ncclResult_t r = m->rccl.GroupStart();
        for (int g = 0; g < G && r == ncclSuccess; g++) {
            unsigned char* b = reinterpret_cast<unsigned char*>(buf[g]);
            r = m->rccl.AllGather(b + (size_t)G * pbytes, b, ZL_PARTIAL_WORDS, ncclUint64, m->comms[g], m->ctx[g]->stream);
        }
        const ncclResult_t r2 = m->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { m->last_rccl = (int)r; drain_all(m); return ZL_EHIP; }
    } else {
        for (int g = 0; g < G; g++) {  // rank g pulls every rank's contribution once that rank has staged it
            hipStream_t st = m->ctx[g]->stream;
            ZL_MHIP(m, m->ctx[g], hipSetDevice(m->dev[g]));
            for (int s = 0; s < G; s++) {
                ZL_MHIP(m, m->ctx[g], hipStreamWaitEvent(st, m->ev[s], 0));
                // hipMemcpyDefault: the two buffers may live on different devices when only SOME ids repeat (unified addressing resolves them)
                ZL_MHIP(m, m->ctx[g], hipMemcpyAsync(reinterpret_cast<unsigned char*>(buf[g]) + (size_t)s * pbytes,
                                                     reinterpret_cast<unsigned char*>(buf[s]) + (size_t)G * pbytes, pbytes, hipMemcpyDefault, st));
            }
        }
    }
    // 3. fold rank 0's gathered copy (every rank holds the same G partials)
    ZL_MHIP(m, m->ctx[0], hipSetDevice(m->dev[0]));
    ZL_MHIP(m, m->ctx[0], hipMemcpyAsync(gathered.data(), buf[0], (size_t)G * pbytes, hipMemcpyDeviceToHost, m->ctx[0]->stream));
    for (int g = 0; g < G; g++) {
        ZL_MHIP(m, m->ctx[g], hipSetDevice(m->dev[g]));
        ZL_MHIP(m, m->ctx[g], hipStreamSynchronize(m->ctx[g]->stream));
    }
    return zl_partials_sum((zl_curve_t)curve, (zl_group_t)group, gathered.data(), (size_t)G, out_xy, out_inf);
}

int zl_ntt_sharded(zl_mctx* m, zl_curve_t curve, void* const* d_data, unsigned log_n, unsigned flags) {
    if (!m || !d_data) return ZL_EINVAL;
    if (flags & ~(ZL_MONT | ZL_COSET | ZL_INVERSE)) return ZL_EINVAL;
    const int G = m->n;
    unsigned log_g = 0;
    while ((1 << log_g) < G) log_g++;
    if ((1 << log_g) != G || log_g < 1 || log_g > 4 || 2 * log_g > log_n) return ZL_EINVAL;
    const unsigned log_m = log_n - log_g;
    const size_t M = (size_t)1 << log_m, B = M >> log_g, chunk = B * 32;
    const bool inverse = (flags & ZL_INVERSE) != 0, mont = (flags & ZL_MONT) != 0;
    const unsigned base = flags & (ZL_INVERSE | ZL_COSET), plain = flags & ZL_INVERSE;
    std::vector<void*> buf(G);
    int rc;
    for (int g = 0; g < G; g++) {
        if (!d_data[g]) return ZL_EINVAL;
        if ((rc = mctx_buf(m, g, M * 32, &buf[g]))) return rc;
    }
    // leg 1 on every rank: forward = cross-rank G-point transform + twiddles (block-column data), inverse = local M-point transform
    rc = per_rank(m, [&](int g) -> int {
        int r = inverse ? zl_ntt_dev(m->ctx[g], curve, d_data[g], log_m, plain | (mont ? ZL_MONT : ZL_MONT_OUT))
                        : zl_ntt_cross_dev(m->ctx[g], curve, d_data[g], log_n, log_g, (unsigned)g, base | (mont ? ZL_MONT : ZL_MONT_OUT));
        if (!r && m->virt) ZL_HIP(m->ctx[g], hipEventRecord(m->ev[g], m->ctx[g]->stream));
        return r;
    });
    if (rc) { drain_all(m); return rc; }
    // the ONE exchange: rank g's chunk j goes to rank j's slot g (all-to-all of G chunks of B elements)
    if (!m->virt) {
        ncclResult_t r = m->rccl.GroupStart();
        for (int g = 0; g < G && r == ncclSuccess; g++) {
            unsigned char* src = reinterpret_cast<unsigned char*>(d_data[g]);
            unsigned char* dst = reinterpret_cast<unsigned char*>(buf[g]);
            for (int j = 0; j < G && r == ncclSuccess; j++) {
                r = m->rccl.Send(src + (size_t)j * chunk, chunk / 8, ncclUint64, j, m->comms[g], m->ctx[g]->stream);
                if (r == ncclSuccess) r = m->rccl.Recv(dst + (size_t)j * chunk, chunk / 8, ncclUint64, j, m->comms[g], m->ctx[g]->stream);
            }
        }
        const ncclResult_t r2 = m->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) { m->last_rccl = (int)r; drain_all(m); return ZL_EHIP; }
    } else {
        for (int g = 0; g < G; g++) {  // receiver g pulls its chunk from every sender once the sender's first leg is done
            hipStream_t st = m->ctx[g]->stream;
            ZL_MHIP(m, m->ctx[g], hipSetDevice(m->dev[g]));
            for (int s = 0; s < G; s++) {
                ZL_MHIP(m, m->ctx[g], hipStreamWaitEvent(st, m->ev[s], 0));
                ZL_MHIP(m, m->ctx[g], hipMemcpyAsync(reinterpret_cast<unsigned char*>(buf[g]) + (size_t)s * chunk,
                                                     reinterpret_cast<const unsigned char*>(d_data[s]) + (size_t)g * chunk, chunk, hipMemcpyDefault, st));
            }
        }
        // a sender's buffer is overwritten by its own second leg: every receiver must have pulled from it first
        for (int g = 0; g < G; g++) {
            ZL_MHIP(m, m->ctx[g], hipSetDevice(m->dev[g]));
            ZL_MHIP(m, m->ctx[g], hipStreamSynchronize(m->ctx[g]->stream));
        }
    }
    // leg 2 on the received data, result copied back into the caller's buffer
    rc = per_rank(m, [&](int g) -> int {
        int r = inverse ? zl_ntt_cross_dev(m->ctx[g], curve, buf[g], log_n, log_g, (unsigned)g, base | (mont ? ZL_MONT : ZL_MONT_IN))
                        : zl_ntt_dev(m->ctx[g], curve, buf[g], log_m, plain | (mont ? ZL_MONT : ZL_MONT_IN));
        if (r) return r;
        ZL_HIP(m->ctx[g], hipMemcpyAsync(d_data[g], buf[g], M * 32, hipMemcpyDeviceToDevice, m->ctx[g]->stream));
        ZL_HIP(m->ctx[g], hipStreamSynchronize(m->ctx[g]->stream));
        return (int)ZL_OK;
    });
    if (rc) drain_all(m);
    return rc;
}

int zl_mctx_last_rccl_error(const zl_mctx* m) { return m ? m->last_rccl : 0; }

}  // extern "C"
