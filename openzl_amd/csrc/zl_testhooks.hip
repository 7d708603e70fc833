// zl_testhooks.hip -- the TEST-ONLY hooks declared in include/zl_backend_test.h: device Poseidon known-answer run, raw-limb access to the
// lazily reduced 28-bit field and to the point formulas (host and device).  Nothing in the product path calls into this file.
#include <string.h>
#include <type_traits>
#include <vector>
#include "../../include/zl_backend_test.h"
#include "zl_ctx.h"
#include "zl_host.h"
#include "zl_field28r.h"

using namespace openzl;

// ------------------------------------------------------------------------------------------------ Poseidon on the device
// One wavefront; every lane runs the whole width-3 permutation (schedule: openzl-tutorials/src/poseidon.rs:165-222) with the device
// Fr arithmetic: canonical -> Montgomery conversion of the state AND of the round constants, the Cauchy MDS entries 1 / (i + 3 + j)
// by Fermat inversion, 63 rounds of add / x^5 / MDS, Montgomery -> canonical.  Only the LFSR bit stream comes from the host.
template <class FrP>
__global__ void __launch_bounds__(64) k_test_poseidon(const uint32_t* __restrict__ keys_canon, int full_rounds, int partial_rounds, uint32_t* __restrict__ state,
                                                       uint32_t* __restrict__ disagree) {
    using F = Fp<FrP>;
    F st[3], mds[3][3];
    for (int i = 0; i < 3; i++) {
        F c;
        for (int k = 0; k < FrP::N; k++) c.l[k] = state[i * FrP::N + k];
        st[i] = zl::to_mont(c);
        for (int j = 0; j < 3; j++) mds[i][j] = zl::inv(zl::from_u64<FrP>((uint64_t)(i + 3 + j)));
    }
    const int half = full_rounds / 2;
    for (int rnd = 0; rnd < full_rounds + partial_rounds; rnd++) {
        for (int i = 0; i < 3; i++) {
            F c;
            for (int k = 0; k < FrP::N; k++) c.l[k] = keys_canon[(3 * rnd + i) * FrP::N + k];
            st[i] = zl::add(st[i], zl::to_mont(c));
        }
        const int lanes = (rnd < half || rnd >= half + partial_rounds) ? 3 : 1;
        for (int i = 0; i < lanes; i++) {
            const F x2 = zl::sqr(st[i]), x4 = zl::sqr(x2);
            st[i] = zl::mul(x4, st[i]);
        }
        F nx[3];
        for (int i = 0; i < 3; i++) {
            F acc = zl::mul(mds[i][0], st[0]);
            acc = zl::add(acc, zl::mul(mds[i][1], st[1]));
            nx[i] = zl::add(acc, zl::mul(mds[i][2], st[2]));
        }
        for (int i = 0; i < 3; i++) st[i] = nx[i];
    }
    uint32_t bad = 0;
    for (int i = 0; i < 3; i++) {
        const F c = zl::from_mont(st[i]);
        for (int k = 0; k < FrP::N; k++) {
            const uint32_t v0 = __shfl(c.l[k], 0);
            bad |= v0 ^ c.l[k];
        }
        if (threadIdx.x == 0) for (int k = 0; k < FrP::N; k++) state[i * FrP::N + k] = c.l[k];
    }
    if (bad) atomicOr(disagree, 1u);
}

// The same permutation on the lazily reduced 10 x 28-bit Fr of the NTT passes (zl_field28r.h, round 4): since that round the hot NTT multiplies with
// mul28r_asm, not with the 8 x 32 carry chain above, so the one reference-held vector is run through THIS multiplier too.
// Everything stays in the multiplier's own Montgomery form x R' (R' = 2^280): to_mont = mul(x, R'^2), products mul(a R', b R') = a b R', lazy additions
// (bounds: an MDS row sum of three products < 6r, plus a round key < 8r; 8 * 8 = 64 <= MUL_BOUND of either instance: 70 at 9 x 29 bits, 2^25 at 10 x 28), Fermat inversion for the Cauchy entries with the same
// multiplier, from_mont = mul(x R', 1), canon, pack.  No value is ever compared with r before the final canon.
template <class FrP, class P28>
__global__ void __launch_bounds__(64) k_test_poseidon28r(const uint32_t* __restrict__ keys_canon, int full_rounds, int partial_rounds, uint32_t* __restrict__ state,
                                                          uint32_t* __restrict__ disagree) {
    using E = Fr28<P28>;
    uint32_t w[8];
    for (int k = 0; k < 8; k++) w[k] = P28::rp2(k);
    const E rp2 = zl::unpack28r<P28>(w);
    for (int k = 0; k < 8; k++) w[k] = k == 0 ? 1u : 0u;
    const E plain_one = zl::unpack28r<P28>(w);
    const E one_m = zl::mul(plain_one, rp2);  // R' mod r (< 2r)
    // exponent r - 2 from the 32-bit-word modulus
    uint32_t e[8];
    {
        uint32_t borrow = 2;
        for (int i = 0; i < 8; i++) {
            const uint32_t m = FrP::mod(i);
            e[i] = m - borrow;
            borrow = m < borrow ? 1u : 0u;
        }
    }
    E st[3], mds[3][3];
    for (int i = 0; i < 3; i++) {
        for (int k = 0; k < 8; k++) w[k] = state[i * 8 + k];
        st[i] = zl::mul(zl::unpack28r<P28>(w), rp2);
        for (int j = 0; j < 3; j++) {
            for (int k = 0; k < 8; k++) w[k] = k == 0 ? (uint32_t)(i + 3 + j) : 0u;
            const E a = zl::mul(zl::unpack28r<P28>(w), rp2);
            E acc = one_m;
            for (int b = 255; b >= 0; b--) {
                acc = zl::mul(acc, acc);
                if ((e[b >> 5] >> (b & 31)) & 1) acc = zl::mul(acc, a);
            }
            mds[i][j] = acc;
        }
    }
    const int half = full_rounds / 2;
    for (int rnd = 0; rnd < full_rounds + partial_rounds; rnd++) {
        for (int i = 0; i < 3; i++) {
            for (int k = 0; k < 8; k++) w[k] = keys_canon[(3 * rnd + i) * 8 + k];
            st[i] = zl::add(st[i], zl::mul(zl::unpack28r<P28>(w), rp2));  // < 6r + 2r
        }
        const int lanes = (rnd < half || rnd >= half + partial_rounds) ? 3 : 1;
        for (int i = 0; i < lanes; i++) {
            const E x2 = zl::mul(st[i], st[i]), x4 = zl::mul(x2, x2);
            st[i] = zl::mul(x4, st[i]);
        }
        E nx[3];
        for (int i = 0; i < 3; i++) {
            E acc = zl::mul(mds[i][0], st[0]);
            acc = zl::add(acc, zl::mul(mds[i][1], st[1]));
            nx[i] = zl::add(acc, zl::mul(mds[i][2], st[2]));  // < 6r
        }
        for (int i = 0; i < 3; i++) st[i] = nx[i];
    }
    uint32_t bad = 0;
    for (int i = 0; i < 3; i++) {
        const E c = zl::canon(zl::mul(st[i], plain_one));
        zl::pack28r<P28>(w, c);
        for (int k = 0; k < 8; k++) {
            const uint32_t v0 = __shfl(w[k], 0);
            bad |= v0 ^ w[k];
        }
        if (threadIdx.x == 0) for (int k = 0; k < 8; k++) state[i * 8 + k] = w[k];
    }
    if (bad) atomicOr(disagree, 1u);
}

template <class FrP, class P28 = void>
static int poseidon_dev_t(zl_ctx* ctx, uint64_t* state) {
    using C = poseidon::Constants<FrP>;
    static const C cst;  // host mirror: Grain LFSR stream -> canonical integers are recovered below
    std::vector<uint32_t> keys(cst.round_keys.size() * FrP::N);
    for (size_t i = 0; i < cst.round_keys.size(); i++) {
        const Fp<FrP> c = zl::from_mont(cst.round_keys[i]);  // back to the LFSR's integer: the device redoes the conversion
        memcpy(&keys[i * FrP::N], c.l, FrP::N * 4);
    }
    void* d = nullptr;
    const size_t kb = keys.size() * 4, sb = 3 * FrP::N * 4;
    int rc = zl_scratch_get(ctx, 9, kb + sb + 64, &d);
    if (rc) return rc;
    uint32_t* d_keys = (uint32_t*)d;
    uint32_t* d_state = d_keys + keys.size();
    uint32_t* d_bad = d_state + 3 * FrP::N;
    hipStream_t st = ctx->stream;
    ZL_HIP(ctx, hipMemcpyAsync(d_keys, keys.data(), kb, hipMemcpyHostToDevice, st));
    ZL_HIP(ctx, hipMemcpyAsync(d_state, state, sb, hipMemcpyHostToDevice, st));
    ZL_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, st));
    if constexpr (std::is_void<P28>::value) hipLaunchKernelGGL((k_test_poseidon<FrP>), dim3(1), dim3(64), 0, st, d_keys, C::FULL_ROUNDS, C::PARTIAL_ROUNDS, d_state, d_bad);
    else hipLaunchKernelGGL((k_test_poseidon28r<FrP, P28>), dim3(1), dim3(64), 0, st, d_keys, C::FULL_ROUNDS, C::PARTIAL_ROUNDS, d_state, d_bad);
    ZL_HIP(ctx, hipGetLastError());
    uint32_t bad = 0;
    ZL_HIP(ctx, hipMemcpyAsync(state, d_state, sb, hipMemcpyDeviceToHost, st));
    ZL_HIP(ctx, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, st));
    ZL_HIP(ctx, hipStreamSynchronize(st));
    return bad ? ZL_EHIP : ZL_OK;
}

// ------------------------------------------------------------------------------------------------ raw-limb field / point access
using F28 = BlsG1::F;
static_assert(F28::L == 14, "test hooks are written for the 14-limb field");

template <class F = F28>
ZL_HD static F raw_load(const uint32_t* w) {
    F r = F::zero();
    for (int i = 0; i < F::L; i++) r.l[i] = w[i];
    return r;
}
template <class F = F28>
ZL_HD static void raw_store(uint32_t* w, const F& a) {
    for (int i = 0; i < F::L; i++) w[i] = a.l[i];
}
// one field operation on raw limbs; F = the 14-limb BLS12-381 field (zl_test_fp28_op) or the 10-limb BN254 field (zl_test_fp28_bn_op, round 4)
template <class F>
ZL_HD static void fp28_op(int op, const uint32_t* in, uint32_t* out) {
    constexpr int L = F::L, CW = F::CANON_WORDS;
    const F a = raw_load<F>(in), b = raw_load<F>(in + L), c = raw_load<F>(in + 2 * L), d = raw_load<F>(in + 3 * L);
    F r = F::zero();
    switch (op) {
    case 0: r = zl::mul(a, b); break;
    case 1: r = zl::sqr(a); break;
    case 2: r = zl::muladd(a, b, c, d); break;
    case 3: r = zl::add(a, b); break;
    case 4: r = zl::dbl(a); break;
    case 5: r = zl::subk<1>(a, b); break;
    case 6: r = zl::subk<2>(a, b); break;
    case 7: r = zl::subk<3>(a, b); break;
    case 8: r = zl::subk<4>(a, b); break;
    case 9: r = zl::subk<5>(a, b); break;
    case 10: r = zl::subk<6>(a, b); break;
    case 11: r = zl::wred(a); break;
    case 12: r = zl::canon(a); break;
    case 13: r.l[0] = a.is_zero() ? 1u : 0u; break;
    case 14: r.l[0] = (a == b) ? 1u : 0u; break;
    case 15: r = zl::muladd4(a, b, c, d, a, d, c, b); break;
    case 19: r = zl::muladd(a, zl::subk_scan<4>(b, c), zl::negk_scan<4>(d), a); break;  // a (b - c + 16q) + (16q - d) a: scan-only operands (un-carried on the device)
    case 20: r = zl::mul(zl::negk_scan<2>(a), b); break;                                  // (4q - a) b
    case 21: r = zl::x3_of(a, b, c); break;                                               // a - b - 2c + 6q in one pass (device), b, c carried
    case 17: r = FieldIO<F>::load_canon(in); break;
    case 18: {
        uint32_t w[CW];
        FieldIO<F>::store_canon(w, a);
        for (int i = 0; i < CW; i++) r.l[i] = w[i];
        break;
    }
    default: break;
    }
    raw_store<F>(out, r);
}
template <class F>
static __global__ void __launch_bounds__(64) k_test_fp28(int op, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fp28_op<F>(op, in + (size_t)i * 4 * F::L, out + (size_t)i * F::L);
}


// ------------------------------------------------------------------------------------------------ the lazy 10 x 28-bit scalar field (zl_field28r.h)
// in: records of two 8-word values a, b (< 2^256); out: 8 words (the canonical result, packed).  `j` selects the bias 2^j r of the subtractions.
//   0 canon(mul(a, b))   1 canon(add(a, b))   2 canon(subk(a, b, 2^j r))   3 canon(a)
//   4 a lazy chain at the bounds the NTT passes reach: x = a; four times { u = x + x; v = x - b' + 2^j r; x = u + v } with b' = mul(b, b), then canon(mul(x, b))
template <class P>
ZL_HD static void fr28_op(int op, int j, const uint32_t* in, uint32_t* out) {
    using E = Fr28<P>;
    const E a = zl::unpack28r<P>(in), b = zl::unpack28r<P>(in + 8);
    uint32_t K[P::L];
    for (int i = 0; i < P::L; i++) K[i] = P::kq(j, i);
    E r = a;
    switch (op) {
    case 0: r = zl::canon(zl::mul(a, b)); break;
    case 1: r = zl::canon(zl::add(a, b)); break;
    case 2: r = zl::canon(zl::subk(a, b, K)); break;
    case 3: r = zl::canon(a); break;
    case 4: {
        const E bb = zl::mul(b, b);  // < 2r
        E x = a;
        for (int k = 0; k < 4; k++) {
            const E u = zl::add(x, x), v = zl::subk(x, bb, K);
            x = zl::add(u, v);
            if (P::MUL_BOUND < 1024) x = zl::wred(x);  // the 29-bit instance: 3 x + 2^j r stays below 128 r, the weak reduction brings it back below 2 r (as the passes do every round)
        }
        r = zl::canon(zl::mul(x, b));
        break;
    }
    case 5: r = zl::canon(zl::wred(zl::add(zl::add(a, a), zl::add(b, b)))); break;  // wred at the top of its range: 2a + 2b < 2^258
    default: break;
    }
    zl::pack28r<P>(out, r);
}
template <class P>
static __global__ void __launch_bounds__(64) k_test_fr28(int op, int j, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr28_op<P>(op, j, in + (size_t)i * 16, out + (size_t)i * 8);
}

template <class PBls, class PBn>
static int test_fr_lazy_op_t(zl_ctx* ctx, zl_curve_t curve, int op, int j, const uint32_t* in, size_t n, uint32_t* out) {
    if ((!in || !out) && n) return ZL_EINVAL;
    if (op < 0 || op > 5 || j < 0 || j > PBls::KQ_MAX || n >= (1u << 24) || (curve != ZL_BLS12_381 && curve != ZL_BN254)) return ZL_EINVAL;
    if (!n) return ZL_OK;
    if (!ctx) {
        for (size_t i = 0; i < n; i++) {
            if (curve == ZL_BLS12_381) fr28_op<PBls>(op, j, in + i * 16, out + i * 8);
            else fr28_op<PBn>(op, j, in + i * 16, out + i * 8);
        }
        return ZL_OK;
    }
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return run_dev(ctx, in, n * 16, out, n * 8, [&](const uint32_t* d_in, uint32_t* d_out, hipStream_t st) {
        const dim3 grid((uint32_t)((n + 63) / 64)), block(64);
        if (curve == ZL_BLS12_381) hipLaunchKernelGGL((k_test_fr28<PBls>), grid, block, 0, st, op, j, d_in, (uint32_t)n, d_out);
        else hipLaunchKernelGGL((k_test_fr28<PBn>), grid, block, 0, st, op, j, d_in, (uint32_t)n, d_out);
    });
}

template <class F> struct RawIO;
template <> struct RawIO<F28> {
    static constexpr int W = 14;
    ZL_HD static F28 load(const uint32_t* w) { return raw_load(w); }
    ZL_HD static void store(uint32_t* w, const F28& a) { raw_store(w, a); }
};
template <bool I> struct RawIO<Fp2LT<F28, I>> {
    static constexpr int W = 28;
    ZL_HD static Fp2LT<F28, I> load(const uint32_t* w) { return Fp2LT<F28, I>{raw_load(w), raw_load(w + 14)}; }
    ZL_HD static void store(uint32_t* w, const Fp2LT<F28, I>& a) { raw_store(w, a.c0); raw_store(w + 14, a.c1); }
};
template <class F>
ZL_HD static void point_op(int op, const uint32_t* in, uint32_t* out) {
    constexpr int W = RawIO<F>::W;
    XYZZ<F> p{RawIO<F>::load(in), RawIO<F>::load(in + W), RawIO<F>::load(in + 2 * W), RawIO<F>::load(in + 3 * W)};
    const XYZZ<F> q{RawIO<F>::load(in + 4 * W), RawIO<F>::load(in + 5 * W), RawIO<F>::load(in + 6 * W), RawIO<F>::load(in + 7 * W)};
    switch (op) {
    case 0: zl::add_mixed(p, q.x, q.y, false); break;
    case 1: zl::add_mixed(p, q.x, q.y, true); break;
    case 2: zl::add_full(p, q); break;
    case 3: zl::dbl_inplace(p); break;
    case 4: p = zl::dbl_affine(p.x, p.y); break;
    case 5: zl::neg_inplace(p); break;
    case 6: {
        const Affine<F> a = zl::to_affine(p);
        p.x = a.x; p.y = a.y; p.zz = F::one(); p.zzz = F::one();
        if (a.is_inf()) { p.zz = F::zero(); p.zzz = F::zero(); }
        break;
    }
    default: break;
    }
    RawIO<F>::store(out, p.x);
    RawIO<F>::store(out + W, p.y);
    RawIO<F>::store(out + 2 * W, p.zz);
    RawIO<F>::store(out + 3 * W, p.zzz);
}
template <class F>
static __global__ void __launch_bounds__(64) k_test_point(int op, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = RawIO<F>::W;
    point_op<F>(op, in + (size_t)i * 8 * W, out + (size_t)i * 4 * W);
}

template <class Launch>
static int run_dev(zl_ctx* ctx, const uint32_t* in, size_t in_words, uint32_t* out, size_t out_words, Launch launch) {
    void* d = nullptr;
    int rc = zl_scratch_get(ctx, 9, (in_words + out_words) * 4 + 64, &d);
    if (rc) return rc;
    uint32_t* d_in = (uint32_t*)d;
    uint32_t* d_out = d_in + in_words;
    hipStream_t st = ctx->stream;
    ZL_HIP(ctx, hipMemcpyAsync(d_in, in, in_words * 4, hipMemcpyHostToDevice, st));
    launch(d_in, d_out, st);
    ZL_HIP(ctx, hipGetLastError());
    ZL_HIP(ctx, hipMemcpyAsync(out, d_out, out_words * 4, hipMemcpyDeviceToHost, st));
    ZL_HIP(ctx, hipStreamSynchronize(st));
    return ZL_OK;
}


// ------------------------------------------------------------------------------------------------ live-data multiplier rate
// The ceiling bench.py's integer-ALU roofline is quoted against, measured on the box of the run: every lane chains x <- x * y with the product scan
// of the accumulation kernel (zl_mul28_gfx950.h) on its OWN pseudo-random operands.  Constant-pattern operands (hipMemset, as tools/fbench28_asm.hip
// uses) run 11 % faster on MI355X -- the chip clocks to its power budget and identical lanes toggle less -- and a launch of a few milliseconds after an
// idle gap runs 10-15 % slower than the steady state (clock ramp): callers warm up and time launches of >= 0.1 s (profiles/r04_fbench_f64.log).
__global__ void __launch_bounds__(64) k_test_mul_rate(uint32_t* __restrict__ sink, int iters, unsigned long long* __restrict__ clk = nullptr) {
    using A = BLS12_381_Fq28;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = 0x9E3779B97F4A7C15ull * (t + 1);
    F28 x = F28::zero(), y = F28::zero();
    for (int k = 0; k < 14; k++) {
        s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        x.l[k] = (uint32_t)s & 0xFFFFFFFu;
        y.l[k] = (uint32_t)(s >> 32) & 0xFFFFFFFu;
    }
    x.l[13] %= A::mod(13);
    y.l[13] %= A::mod(13);
    // clk != nullptr: shader-cycle (s_memtime) and 100-MHz (s_memrealtime) counters around the chain, one record per wave (zl_test_fq_mul_clock)
    unsigned long long c0 = 0, w0 = 0;
    if (clk) { c0 = __builtin_readcyclecounter(); w0 = __builtin_amdgcn_s_memrealtime(); }
    for (int k = 0; k < iters; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
        F28 r = x;
        mul28_asm<A>(r.l, x.l, y.l);
        x = r;
#endif
    }
    if (clk) {
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) { clk[4 * blockIdx.x] = c0; clk[4 * blockIdx.x + 1] = c1; clk[4 * blockIdx.x + 2] = w0; clk[4 * blockIdx.x + 3] = w1; }
    }
    uint32_t acc = 0;
    for (int k = 0; k < 14; k++) acc ^= x.l[k];
    sink[t] = acc;
}

template <class F>
static int test_fp28_op_t(zl_ctx* ctx, int op, const uint32_t* in, size_t n, uint32_t* out) {
    if ((!in || !out) && n) return ZL_EINVAL;
    if (op < 0 || op > 21 || op == 16 || n >= (1u << 24)) return ZL_EINVAL;
    if (!n) return ZL_OK;
    constexpr size_t L = F::L;
    if (!ctx) {
        for (size_t i = 0; i < n; i++) fp28_op<F>(op, in + i * 4 * L, out + i * L);
        return ZL_OK;
    }
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return run_dev(ctx, in, n * 4 * L, out, n * L, [&](const uint32_t* d_in, uint32_t* d_out, hipStream_t st) {
        hipLaunchKernelGGL((k_test_fp28<F>), dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, st, op, d_in, (uint32_t)n, d_out);
    });
}
extern "C" {

int zl_test_poseidon_permute_dev(zl_ctx* ctx, zl_curve_t curve, uint64_t* state) {
    if (!ctx || !state) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    if (curve == ZL_BLS12_381) return poseidon_dev_t<BLS12_381_Fr>(ctx, state);
    if (curve == ZL_BN254) return poseidon_dev_t<BN254_Fr>(ctx, state);
    return ZL_EINVAL;
}

int zl_test_poseidon_permute_dev28r(zl_ctx* ctx, zl_curve_t curve, uint64_t* state) {
    if (!ctx || !state) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    // both instances of the lazy field: nine 29-bit limbs (what the NTT passes multiply with since round 5) and ten 28-bit limbs (round 4, kept as the A/B build);
    // the 29-bit result is returned, a disagreement between the two is an error
    uint64_t s28[12];
    memcpy(s28, state, sizeof s28);
    int rc = curve == ZL_BLS12_381 ? poseidon_dev_t<BLS12_381_Fr, BLS12_381_Fr29>(ctx, state) : poseidon_dev_t<BN254_Fr, BN254_Fr29>(ctx, state);
    if (rc) return rc;
    rc = curve == ZL_BLS12_381 ? poseidon_dev_t<BLS12_381_Fr, BLS12_381_Fr28>(ctx, s28) : poseidon_dev_t<BN254_Fr, BN254_Fr28>(ctx, s28);
    if (rc) return rc;
    return memcmp(s28, state, sizeof s28) ? ZL_EHIP : ZL_OK;
}

int zl_test_fp28_op(zl_ctx* ctx, int op, const uint32_t* in, size_t n, uint32_t* out) { return test_fp28_op_t<F28>(ctx, op, in, n, out); }
#ifndef ZL_BN_FIELD32
int zl_test_fp28_bn_op(zl_ctx* ctx, int op, const uint32_t* in, size_t n, uint32_t* out) { return test_fp28_op_t<BnG1::F>(ctx, op, in, n, out); }
#else
int zl_test_fp28_bn_op(zl_ctx*, int, const uint32_t*, size_t, uint32_t*) { return ZL_EINVAL; }  // (the A/B build keeps BN254 on the 32-bit field)
#endif

int zl_test_point_op(zl_ctx* ctx, zl_group_t group, int hot, int op, const uint32_t* in, size_t n, uint32_t* out) {
    if ((!in || !out) && n) return ZL_EINVAL;
    if (op < 0 || op > 6 || n >= (1u << 22) || (group != ZL_G1 && group != ZL_G2)) return ZL_EINVAL;
    if (!n) return ZL_OK;
    const size_t W = group == ZL_G1 ? 14 : 28;
    if (!ctx) {
        for (size_t i = 0; i < n; i++) {
            if (group == ZL_G1) point_op<F28>(op, in + i * 8 * W, out + i * 4 * W);
            else if (hot) point_op<Fp2LT<F28, true>>(op, in + i * 8 * W, out + i * 4 * W);
            else point_op<Fp2LT<F28, false>>(op, in + i * 8 * W, out + i * 4 * W);
        }
        return ZL_OK;
    }
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    return run_dev(ctx, in, n * 8 * W, out, n * 4 * W, [&](const uint32_t* d_in, uint32_t* d_out, hipStream_t st) {
        const dim3 grid((uint32_t)((n + 63) / 64)), block(64);
        if (group == ZL_G1) hipLaunchKernelGGL((k_test_point<F28>), grid, block, 0, st, op, d_in, (uint32_t)n, d_out);
        else if (hot) hipLaunchKernelGGL((k_test_point<Fp2LT<F28, true>>), grid, block, 0, st, op, d_in, (uint32_t)n, d_out);
        else hipLaunchKernelGGL((k_test_point<Fp2LT<F28, false>>), grid, block, 0, st, op, d_in, (uint32_t)n, d_out);
    });
}

int zl_test_fr28_op(zl_ctx* ctx, zl_curve_t curve, int op, int j, const uint32_t* in, size_t n, uint32_t* out) {
    if ((!in || !out) && n) return ZL_EINVAL;
    return test_fr_lazy_op_t<BLS12_381_Fr28, BN254_Fr28>(ctx, curve, op, j, in, n, out);
}
int zl_test_fr29_op(zl_ctx* ctx, zl_curve_t curve, int op, int j, const uint32_t* in, size_t n, uint32_t* out) {
    return test_fr_lazy_op_t<BLS12_381_Fr29, BN254_Fr29>(ctx, curve, op, j, in, n, out);
}

int zl_test_fq_mul_rate(zl_ctx* ctx, int waves_per_simd, int iters, double* g_products_per_s) {
    if (!ctx || !g_products_per_s || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || iters > (1 << 20)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t blocks = (uint32_t)ctx->cu_count * 4u * (uint32_t)waves_per_simd;
    void* d = nullptr;
    int rc = zl_scratch_get(ctx, 9, (size_t)blocks * 64 * 4, &d);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    ZL_HIP(ctx, hipEventCreate(&e0));
    ZL_HIP(ctx, hipEventCreate(&e1));
    hipLaunchKernelGGL(k_test_mul_rate, dim3(blocks), dim3(64), 0, st, (uint32_t*)d, 8, (unsigned long long*)nullptr);
    ZL_HIP(ctx, hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_test_mul_rate, dim3(blocks), dim3(64), 0, st, (uint32_t*)d, iters, (unsigned long long*)nullptr);
    ZL_HIP(ctx, hipEventRecord(e1, st));
    ZL_HIP(ctx, hipStreamSynchronize(st));
    ZL_HIP(ctx, hipGetLastError());
    float ms = 0;
    ZL_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *g_products_per_s = (double)blocks * 64.0 * (double)iters / ((double)ms * 1e-3) / 1e9;
    return ZL_OK;
}

// Reduces `waves` records {cycles at start, cycles at end, 100-MHz ticks at start, ticks at end} (one per wave) to
// out[0] effective shader clock in GHz = sum of cycle deltas / sum of tick deltas * 0.1, out[1] span of the launch in ms by the tick counter (last end - first start),
// out[2] waves, out[3] mean life of a wave in ms, out[4] / out[5] smallest / largest per-wave clock in GHz
static void clock_reduce(const std::vector<unsigned long long>& r, size_t waves, double* out) {
    double sc = 0, sw = 0, lo = 1e30, hi = 0;
    unsigned long long wmin = ~0ull, wmax = 0;
    size_t used = 0;
    for (size_t i = 0; i < waves; i++) {
        const unsigned long long c0 = r[4 * i], c1 = r[4 * i + 1], w0 = r[4 * i + 2], w1 = r[4 * i + 3];
        if (w1 <= w0 || c1 <= c0) continue;  // (a wave whose lanes all returned at once)
        used++;
        sc += (double)(c1 - c0);
        sw += (double)(w1 - w0);
        const double g = (double)(c1 - c0) / (double)(w1 - w0) * 0.1;
        if (w1 - w0 > 1000) { lo = g < lo ? g : lo; hi = g > hi ? g : hi; }  // per-wave extremes only over waves that lived > 10 us (tick granularity)
        wmin = w0 < wmin ? w0 : wmin;
        wmax = w1 > wmax ? w1 : wmax;
    }
    out[0] = sw > 0 ? sc / sw * 0.1 : 0;
    out[1] = used ? (double)(wmax - wmin) / 1e5 : 0;
    out[2] = (double)used;
    out[3] = used ? sw / (double)used / 1e5 : 0;
    out[4] = lo > 1e29 ? 0 : lo;
    out[5] = hi;
}

int zl_test_fq_mul_clock(zl_ctx* ctx, int waves_per_simd, int iters, double* out) {
    if (!ctx || !out || waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || iters > (1 << 20)) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t blocks = (uint32_t)ctx->cu_count * 4u * (uint32_t)waves_per_simd;
    void* d = nullptr;
    int rc = zl_scratch_get(ctx, 9, (size_t)blocks * 64 * 4 + (size_t)blocks * 32, &d);
    if (rc) return rc;
    unsigned long long* d_clk = (unsigned long long*)((char*)d + (size_t)blocks * 64 * 4);
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    ZL_HIP(ctx, hipEventCreate(&e0));
    ZL_HIP(ctx, hipEventCreate(&e1));
    ZL_HIP(ctx, hipMemsetAsync(d_clk, 0, (size_t)blocks * 32, st));
    hipLaunchKernelGGL(k_test_mul_rate, dim3(blocks), dim3(64), 0, st, (uint32_t*)d, 8, (unsigned long long*)nullptr);
    ZL_HIP(ctx, hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_test_mul_rate, dim3(blocks), dim3(64), 0, st, (uint32_t*)d, iters, d_clk);
    ZL_HIP(ctx, hipEventRecord(e1, st));
    std::vector<unsigned long long> r((size_t)blocks * 4);
    ZL_HIP(ctx, hipMemcpyAsync(r.data(), d_clk, (size_t)blocks * 32, hipMemcpyDeviceToHost, st));
    ZL_HIP(ctx, hipStreamSynchronize(st));
    ZL_HIP(ctx, hipGetLastError());
    float ms = 0;
    ZL_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    clock_reduce(r, blocks, out);
    out[6] = (double)blocks * 64.0 * (double)iters / ((double)ms * 1e-3) / 1e9;
    out[7] = (double)ms;
    return ZL_OK;
}

// A clock probe for kernels that carry no instrumentation (the NTT passes): eight one-wave blocks (one per XCD by the dispatcher's round robin) spin on the
// 100-MHz counter for `spin_us` microseconds, sleeping between reads, on a stream of their own; each leaves its cycle and tick deltas.  Launched right BEFORE
// the work to be observed, read after it: the effective clock of the chip while that work ran (the spin itself is one sleeping wave per XCD).
static __global__ void __launch_bounds__(64) k_clock_spin(unsigned long long* __restrict__ out, unsigned long long ticks) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long w = w0;
    while (w - w0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        w = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[4 * blockIdx.x] = c0; out[4 * blockIdx.x + 1] = c1; out[4 * blockIdx.x + 2] = w0; out[4 * blockIdx.x + 3] = w; }
}
static hipStream_t g_probe_stream = nullptr;
static unsigned long long* g_probe_buf = nullptr;

int zl_test_clock_probe_launch(zl_ctx* ctx, unsigned spin_us) {
    if (!ctx || spin_us < 1 || spin_us > 2000000) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    if (!g_probe_stream) {
        // in the high stream-priority class: those streams have hardware queues of their own, so the probe never shares a queue (= runs in turn) with a chain of
        // the work it is meant to observe (round 6: on a default-class stream the probe of a pipelined MSM batch read 2.40 GHz -- it had run beside nothing)
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        ZL_HIP(ctx, hipStreamCreateWithPriority(&g_probe_stream, hipStreamNonBlocking, hi));
    }
    if (!g_probe_buf) ZL_HIP(ctx, hipMalloc((void**)&g_probe_buf, 8 * 32));
    ZL_HIP(ctx, hipMemsetAsync(g_probe_buf, 0, 8 * 32, g_probe_stream));
    hipLaunchKernelGGL(k_clock_spin, dim3(8), dim3(64), 0, g_probe_stream, g_probe_buf, (unsigned long long)spin_us * 100ull);
    ZL_HIP(ctx, hipGetLastError());
    return ZL_OK;
}

int zl_test_clock_probe_read(zl_ctx* ctx, double* out) {
    if (!ctx || !out || !g_probe_stream || !g_probe_buf) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    ZL_HIP(ctx, hipStreamSynchronize(g_probe_stream));
    std::vector<unsigned long long> r(8 * 4);
    ZL_HIP(ctx, hipMemcpy(r.data(), g_probe_buf, 8 * 32, hipMemcpyDeviceToHost));
    clock_reduce(r, 8, out);
    return ZL_OK;
}

int zl_test_acc_clock(zl_ctx* ctx, int on) {
    if (!ctx) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    ZL_HIP(ctx, hipDeviceSynchronize());
    if (!on) {
        if (ctx->acc_clk) (void)hipFree(ctx->acc_clk);
        ctx->acc_clk = nullptr;
        ctx->acc_clk_cap = ctx->acc_clk_waves = 0;
        return ZL_OK;
    }
    if (!ctx->acc_clk) {
        const size_t cap = (size_t)(1u << 18) * 32;  // 2^18 waves of 64 chunks: 2^24 chunks
        ZL_HIP(ctx, hipMalloc(&ctx->acc_clk, cap));
        ctx->acc_clk_cap = cap;
    }
    ctx->acc_clk_waves = 0;
    return ZL_OK;
}

int zl_test_acc_clock_read(zl_ctx* ctx, double* out) {
    if (!ctx || !out) return ZL_EINVAL;
    if (!ctx->acc_clk || !ctx->acc_clk_waves) return ZL_EINVAL;  // not armed, or no large G1 accumulation ran since
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    ZL_HIP(ctx, hipDeviceSynchronize());
    std::vector<unsigned long long> r(ctx->acc_clk_waves * 4);
    ZL_HIP(ctx, hipMemcpy(r.data(), ctx->acc_clk, r.size() * 8, hipMemcpyDeviceToHost));
    clock_reduce(r, ctx->acc_clk_waves, out);
    return ZL_OK;
}

}  // extern "C"
