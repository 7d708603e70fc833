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
//
// Units: zl_msm_sort.hip (steps 1-2, curve-independent, compiled once), zl_msm_acc.hip (step 3) and zl_msm_tail.hip (steps 4-5) hold the device
// code of the heavy kernels per group; this file is the per-group host side: MsmJob (zl_msm_job.h), the single-call and pipelined drivers,
// the bases handles, and the light kernels they launch (zl_msm_endo.h, zl_msm_bases.h).
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "zl_ctx.h"
#include "zl_pool.h"
#include "zl_quad.h"

#ifndef ZL_G
#error "compile with -DZL_G=<group config>"
#endif
#include "zl_msm_job.h"
#include "zl_msm_bases.h"
// the accumulation and tail kernels of this group are defined in zl_msm_acc.hip / zl_msm_tail.hip
ZL_MSM_ACCUMULATE_KERNELS(extern, ZL_G)
ZL_MSM_TAIL_KERNELS(extern, ZL_G)

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
            job.set_host_buffer(reinterpret_cast<X*>(ctx->pinned));  // (need = the root channels + the scalar-1 sum + 16 bytes: the status words fit behind them)
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
    // one pipeline per ctx at a time (zl_ctx.h): the event pool, the three buffer sets and the side streams belong to this call
    struct Busy {
        std::atomic<int>& f;
        bool mine;
        explicit Busy(std::atomic<int>& x) : f(x), mine(x.exchange(1, std::memory_order_acq_rel) == 0) {}
        ~Busy() { if (mine) f.store(0, std::memory_order_release); }
    } busy(ctx->pipeline_busy);
    if (!busy.mine) return ZL_EINVAL;
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
    // SMALL side-by-side jobs (the MSMs of a small proof: chains of ~25 kernels of 5-20 us) run on the ordinary default-class lanes.  The runtime multiplexes the
    // streams of a process onto a few hardware queues, and two chains on one queue run in turn (tools/queue_chains.hip).  Round 5 tried lanes of their own in the high
    // and in the low stream-priority class and the idle tail streams as lanes: each won in one stream population and lost in another
    // (profiles/r05_small_lanes_*_ab.log); what made the difference was WHEN the streams had come into being, and with every stream of a ctx created at zl_ctx_create
    // (zl_ctx_streams_init) the default-class lanes are the best choice whatever ran before.  The knobs of those experiments were removed in round 6.
    hipStream_t* const lanes = ctx->stream_lane;
    if (side) {
        for (size_t k = 0; k < 4; k++)  // (all four at once, whatever NS: consecutive creations take consecutive queues of the pool)
            if (!lanes[k]) ZL_HIP(ctx, hipStreamCreateWithFlags(&lanes[k], hipStreamNonBlocking));
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
        // grown ONCE to what any ordinary batch needs (24 jobs of 320 root channels: 3.8 MB for G2), not to this call's exact size: a reallocation of pinned memory
        // costs 1-2 ms, and a batch of another length or another window plan paid it inside its own timed region (bench.py's config-2 leg: 3.44 instead of 3.2 ms
        // per MSM for the first 6-step batch behind 3- and 2-step ones, profiles/r06_c2_diag.log)
        const size_t want = std::max(per * count, (sizeof(X) * 320 + 16) * (size_t)24);
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_cap = 0;
        ZL_HIP(ctx, hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_cap = want;
    }
    for (size_t i = 0; i < count; i++) {
        unsigned char* base = reinterpret_cast<unsigned char*>(ctx->pinned) + per * i;
        jobs[i].set_host_buffer(reinterpret_cast<X*>(base));  // (per = the largest job's root channels + scalar-1 sum + 16 bytes: every job's status words fit behind its own)
    }
    hipStream_t s_sort = ctx->stream_sort, s_acc = ctx->stream;
    // The tail of job i runs on the tail stream of its buffer set: consecutive tails are independent (own buckets, partials, tree nodes), and
    // for small jobs -- a chain of ~25 dependent group operations at a few lanes each -- they are what the pipeline's latency consists of:
    // four 237-point MSMs of a small proof finished their tails one after the other in 2.1 ms, now side by side.
    hipStream_t s_tails[3] = {ctx->stream_tail[0], ctx->stream_tail[1], ctx->stream_tail[2]};
    // (A persistent two-wave accumulation that leaves room for the side streams was built in round 3 and removed in round 6: the sort and the tail do move under
    // it, but two co-resident field-arithmetic kernels share one instruction cache -- 49.9 ms per 2^24 MSM against 37.6, profiles/r03_persist_accumulate.log.)
    // (Measured and dropped: making the accumulation of job i+1 wait for the merge kernels / level 0 of job i, so that two field-arithmetic
    // kernels never share the machine -- 958 465-constraint proof 19.2-19.4 ms with or without, 2^20 batches 3.65 = 3.65 ms per MSM.)
    // events from the ctx's pool (zl_ctx_events): [sorted | tail | acc (untimed runs)] without timing, [begin, end | acc0 | acc (timed runs)] with
    hipEvent_t *pool_nt = nullptr, *pool_t = nullptr;
    // (pools grown to at least a 24-job batch at once: an event costs ~50 us to create, and a longer batch than any before paid that inside its own call)
    if ((rc = zl_ctx_events(ctx, 0, 4 * std::max<size_t>(count, 24), &pool_nt))) return rc;
    if ((rc = zl_ctx_events(ctx, 1, 2 + (ctx->timing_on ? 2 * std::max<size_t>(count, 24) : 0), &pool_t))) return rc;
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
        for (size_t k = 0; k < NS && he == hipSuccess; k++) he = hipStreamWaitEvent(lanes[k], ev_begin, 0);
    static const bool jtrace = getenv("ZL_HOST_TRACE") != nullptr;
    const auto jt0 = std::chrono::steady_clock::now();
    // issue of one job: sort | accumulate | tail with the events between them.  pipelined: three phases on three streams; side by side: the
    // whole job on the stream of its buffer set (the waits are then between operations of one stream, i.e. no-ops)
    auto issue_sort = [&](size_t i) -> int {
        hipStream_t js_sort = side ? lanes[i % NS] : s_sort;
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
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        return ZL_OK;
    };
    auto issue_acc_tail = [&](size_t i) -> int {
        hipStream_t js_acc = side ? lanes[i % NS] : s_acc;
        hipStream_t s_tail = side ? lanes[i % NS] : s_tails[i % 3];
        hipError_t e = hipStreamWaitEvent(js_acc, ev_sorted[i], 0);
        if (e == hipSuccess && ctx->timing_on) e = hipEventRecord(ev_acc0[i], js_acc);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return ZL_EHIP; }
        int r;
        if ((r = jobs[i].accumulate(ctx, js_acc))) return r;
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
    auto issue_job = [&](size_t i) -> int {
        const int r = issue_sort(i);
        return r ? r : issue_acc_tail(i);
    };
    // Side by side, LARGE jobs (Groth16's four G1 MSMs of a big circuit): the streams of the lanes share hardware queues (two of them for four lanes
    // with the runtime's default of four queues per process), and a hardware queue runs its packets in order -- job-major issue puts the sort of
    // job 2 behind the accumulation and the tail of job 0 (rocprofv3 of the 958 465-constraint proof, round 4: the l MSM's sort ran 1.5 ms after
    // the machine had gone idle for it).  Phase-major issue instead: the sorts of all jobs that wait for nothing, then their accumulations and
    // tails, then the jobs that wait for an event (the h MSM behind the witness map): 18.1 -> 17.8 ms (profiles/r04_g16_gate_ab.log).  Built on top and
    // removed: the G2 accumulation held back until those sorts had finished (the sort kernels need 104 - 160 registers per SIMD and stall beside the
    // 416-register G2 wave): the a accumulation then starts 3 ms earlier, the G2 accumulation 3 ms later, and the proof takes the same 18.1 ms --
    // the proof is bound by the sum of its group additions, whatever their order.
    const bool phased = side && count <= NS && biggest >= ((uint64_t)1 << zl_tune("ZL_TUNE_PHASED_MIN_LOG", 17));
    // issued[i]: 0 not yet, 1 issued, < 0 failed (-code).  (Round 3 also built one persistent issuing thread per lane: all four jobs of a small proof then reach the
    // device within 0.15 ms, but finish together and later than the staggered jobs of a single issuing thread -- median 1.40 against 1.16 ms per proof; removed in round 6.)
    std::unique_ptr<std::atomic<int>[]> issued(new std::atomic<int>[count]);
    for (size_t i = 0; i < count; i++) issued[i].store(0);
    if (phased) {
        for (size_t i = 0; i < count; i++)
            if (!specs[i].wait && he == hipSuccess && rc == ZL_OK) rc = issue_sort(i);
        for (size_t i = 0; i < count; i++)
            if (!specs[i].wait && he == hipSuccess && rc == ZL_OK) rc = issue_acc_tail(i);
        for (size_t i = 0; i < count; i++)
            if (specs[i].wait && he == hipSuccess && rc == ZL_OK) rc = issue_job(i);
        for (size_t i = 0; i < count; i++) issued[i].store(he == hipSuccess && rc == ZL_OK ? 1 : -(rc ? rc : (int)ZL_EHIP), std::memory_order_release);
    } else {
        for (size_t i = 0; i < count; i++) {
            if (he == hipSuccess && rc == ZL_OK) rc = issue_job(i);
            issued[i].store(he == hipSuccess && rc == ZL_OK ? 1 : -(rc ? rc : (int)ZL_EHIP), std::memory_order_release);
        }
    }
    // Host tails: this thread waits for the jobs' tail events in order (a job's root channels are then in pinned memory) and hands every
    // finished job to a helper thread that runs its window Horner and delivers the result -- while the device works on the later jobs.  Small
    // jobs, which the device finishes faster than the host, get their Horners side by side; `on_done` is delivered in job order.  (The helpers
    // make no HIP calls: a fresh thread's first HIP call costs ~0.1 ms of per-thread runtime setup, per proof.)
    std::atomic<int> frc{ZL_OK};
    size_t delivered = 0;  // in-order delivery of the results: a finisher sleeps on the condition variable until its predecessors have delivered (no spinning:
    std::mutex deliver_mu;  // a batch of many jobs has many finishers alive at once, next to the host pool's workers)
    std::condition_variable deliver_cv;
    std::vector<std::thread> finishers;
    // every job issued (the lanes issue side by side: about the time of one job), then ev_end behind the last tail of every stream that ran tails
    for (size_t i = 0; i < count; i++) {
        int st;
        while ((st = issued[i].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (st < 0 && rc == ZL_OK) rc = -st;
    }
    if (he == hipSuccess && rc == ZL_OK) {
        hipStream_t last_tail = side ? lanes[(count - 1) % NS] : s_tails[(count - 1) % 3];
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
                double values_us = 0;
                if (ok) {
                    const X total = jobs[i].finish(true, jtrace ? &values_us : nullptr);  // (concurrent callers of the host pool each take part in their own loop)
                    memset(out_partials + i * ZL_PARTIAL_WORDS, 0, ZL_PARTIAL_WORDS * 8);
                    memcpy(out_partials + i * ZL_PARTIAL_WORDS, &total, sizeof(X));
                }
                if (jtrace) fprintf(stderr, "[zl_msm jobs] job %zu Horner done at %.1f us (window values %.1f us, %u sets)\n", i,
                                    (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - jt0).count() / 1e3, values_us, (unsigned)jobs[i].SETS);
                {
                    std::unique_lock<std::mutex> lk(deliver_mu);
                    deliver_cv.wait(lk, [&] { return delivered == i; });
                }
                if (ok && on_done && frc.load() == ZL_OK) (*on_done)(i);
                {
                    std::lock_guard<std::mutex> lk(deliver_mu);
                    delivered = i + 1;
                }
                deliver_cv.notify_all();
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
            const hipError_t hs = hipStreamSynchronize(lanes[k]);
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
template <class G>
static int partials_sum_t(const uint64_t* partials, size_t count, uint64_t* out_partial);
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
// one handle holding the concatenation of ranges of other handles of the same group (device-to-device; Groth16's folded C query, zl_groth16.hip)
template <class G>
static int bases_concat_t(zl_ctx* ctx, const zl_bases* const* parts, const size_t* first, const size_t* n, size_t count, zl_bases* out) {
    using F = typename G::F;
    size_t total = 0;
    for (size_t i = 0; i < count; i++) {
        if (!parts[i] || first[i] > parts[i]->n || n[i] > parts[i]->n - first[i]) return ZL_EINVAL;
        total += n[i];
    }
    if (total >= (1ull << 31)) return ZL_EINVAL;
    void* d_pts = nullptr;
    ZL_HIP(ctx, hipMalloc(&d_pts, std::max<size_t>(total, 1) * sizeof(Affine<F>)));
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    size_t at = 0;
    for (size_t i = 0; i < count && e == hipSuccess; i++) {
        if (n[i]) e = hipMemcpyAsync(reinterpret_cast<Affine<F>*>(d_pts) + at, reinterpret_cast<const Affine<F>*>(parts[i]->d_pts) + first[i], n[i] * sizeof(Affine<F>), hipMemcpyDeviceToDevice, st);
        at += n[i];
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ctx->last_hip = (int)e; (void)hipFree(d_pts); return ZL_EHIP; }
    out->d_pts = d_pts;
    out->n = total;
    const int rc = bases_flag_inf_t<G>(ctx, out);
    if (rc) { (void)hipFree(d_pts); out->d_pts = nullptr; out->n = 0; return rc; }
    return ZL_OK;
}
int ZL_GNAME(zl_bases_concat)(zl_ctx* ctx, const zl_bases* const* parts, const size_t* first, const size_t* n, size_t count, zl_bases* out) {
    return bases_concat_t<ZL_G>(ctx, parts, first, n, count, out);
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
