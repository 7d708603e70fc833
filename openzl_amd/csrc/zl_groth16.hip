// zl_groth16.hip -- Groth16 prover on the device: QAP witness map (NTT block) + the five MSMs + host assembly.
//
// Replaces ark_groth16::create_random_proof / create_proof_with_assignment + R1CStoQAP::witness_map (ark-groth16 0.3.0)
// behind `Groth16::<E>::prove` (/root/reference/plugins/arkworks/src/groth16.rs:445-457):
//   a_i = <A_i,z>, b_i = <B_i,z>, c_i = <C_i,z>; a[n_constraints + j] = z[j] for the instance block
//   a,b,c <- coset_fft(ifft(.)); ab = (a o b - c) / (g^N - 1); h = coset_ifft(ab)
//   A = r*delta1 + a_query[0] + MSM(a_query[1..], z[1..]) + alpha1      (B1 in G1, B2 in G2 likewise with s, beta)
//   C = s*A + r*B1 - r*s*delta1 + MSM(l_query, z_wit) + MSM(h_query, h[..N-1])
// The R1CS arrives already synthesised (the reference moves a pre-built constraint system into arkworks the same way,
// plugins/arkworks/src/constraint/mod.rs:179-197); sparse mat-vec, pointwise ops and Montgomery entry/exit are small
// elementwise kernels around zl_ntt_run / zl_msm_run; the window Horner and the final few group operations run on host.
#include <stdlib.h>
#include <string.h>
#include <array>
#include <thread>
#include <vector>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include "zl_ctx.h"

template <class FrP>
__global__ void __launch_bounds__(256) k_fr_to_mont(Fp<FrP>* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = zl::to_mont(v[i]);
}
template <class FrP>
__global__ void __launch_bounds__(256) k_fr_from_mont(const Fp<FrP>* __restrict__ in, Fp<FrP>* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = zl::from_mont(in[i]);
}
// out[row] = sum_k val[k] * z[col[k]]  (Montgomery); rows >= n_rows: out[n_rows + j] = z[j] for j < tail (A only), else 0
template <class FrP>
__global__ void __launch_bounds__(256) k_r1cs_spmv(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ col, const Fp<FrP>* __restrict__ val,
                                                    const Fp<FrP>* __restrict__ z, uint32_t n_rows, uint32_t tail, uint32_t N,
                                                    Fp<FrP>* __restrict__ out) {
    using F = Fp<FrP>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    F acc = F::zero();
    if (i < n_rows) {
        for (uint32_t k = ptr[i]; k < ptr[i + 1]; k++) acc = zl::add(acc, zl::mul(val[k], z[col[k]]));
    } else if (i - n_rows < tail) {
        acc = z[i - n_rows];
    }
    out[i] = acc;
}
// The same product with EIGHT lanes per row (round 6): a row of the Poseidon circuit's A and B holds up to ~150 terms (the linear layers are merged into the rows), and one
// lane per row walks them as a chain of dependent multiply-adds -- 91 + 68 us for the 235 rows of a one-hash proof, on the critical path in front of the h MSM.  Lane j of
// a group takes the terms k = first + j, + 8, ...; the eight partial sums are folded by three xor-shuffle steps (fully reduced Fr: addition is exact in any order).
template <class FrP>
__global__ void __launch_bounds__(256) k_r1cs_spmv8(const uint32_t* __restrict__ ptr, const uint32_t* __restrict__ col, const Fp<FrP>* __restrict__ val,
                                                     const Fp<FrP>* __restrict__ z, uint32_t n_rows, uint32_t tail, uint32_t N,
                                                     Fp<FrP>* __restrict__ out) {
    using F = Fp<FrP>;
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = gt >> 3, j = gt & 7u;
    if (i >= N) return;  // (N is a multiple of 32: whole groups leave together)
    F acc = F::zero();
    if (i < n_rows) {
        const uint32_t e = ptr[i + 1];
        for (uint32_t k = ptr[i] + j; k < e; k += 8) acc = zl::add(acc, zl::mul(val[k], z[col[k]]));
    } else if (j == 0 && i - n_rows < tail) {
        acc = z[i - n_rows];
    }
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
        F o;
#pragma unroll
        for (int w = 0; w < F::N; w++) o.l[w] = (uint32_t)__shfl_xor((int)acc.l[w], d, 8);
        acc = zl::add(acc, o);
    }
    if (j == 0) out[i] = acc;
}
// a = (a*b - c) * zinv
template <class FrP>
__global__ void __launch_bounds__(256) k_qap_pointwise(Fp<FrP>* __restrict__ a, const Fp<FrP>* __restrict__ b, const Fp<FrP>* __restrict__ c,
                                                        Fp<FrP> zinv, uint32_t N) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) a[i] = zl::mul(zl::sub(zl::mul(a[i], b[i]), c[i]), zinv);
}

template <class G>
static XYZZ<typename G::F> affine_from_canon(const uint64_t* xy) {
    using F = typename G::F;
    constexpr int WORDS = FieldIO<F>::WORDS;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(xy);
    uint32_t acc = 0;
    for (int k = 0; k < 2 * WORDS; k++) acc |= w[k];
    if (!acc) return XYZZ<F>::inf();
    return XYZZ<F>::from_affine(Affine<F>{FieldIO<F>::load_canon(w), FieldIO<F>::load_canon(w + WORDS)});
}
template <class G>
static void store_canon(uint64_t* out_xy, uint8_t* out_inf, const XYZZ<typename G::F>& p) {
    using F = typename G::F;
    constexpr int WORDS = FieldIO<F>::WORDS;
    uint32_t* w = reinterpret_cast<uint32_t*>(out_xy);
    *out_inf = p.is_inf() ? 1 : 0;
    if (p.is_inf()) { for (int k = 0; k < 2 * WORDS; k++) w[k] = 0; return; }
    const Affine<F> a = zl::to_affine(p);
    FieldIO<F>::store_canon(w, a.x);
    FieldIO<F>::store_canon(w + WORDS, a.y);
}
template <class F>
static XYZZ<F> from_partial(const uint64_t* partial) {
    XYZZ<F> p;
    memcpy(&p, partial, sizeof p);
    return p;
}

// R1CS matrices -> device (CSR, coefficients converted to Montgomery once).  Static per circuit, like the proving key.
template <class FrP>
static int r1cs_upload_t(zl_ctx* ctx, const zl_r1cs* cs, zl_r1cs_dev* out) {
    using Fr = Fp<FrP>;
    const uint32_t nc = cs->n_constraints;
    size_t bytes = 0;
    auto take = [&](size_t b) { size_t o = bytes; bytes += (b + 255) / 256 * 256; return o; };
    size_t nnz[3];
    for (int m = 0; m < 3; m++) {
        nnz[m] = cs->row_ptr[m][nc];
        out->off_ptr[m] = take((size_t)(nc + 1) * 4);
        out->off_col[m] = take(nnz[m] * 4);
        out->off_val[m] = take(nnz[m] * 32);
        out->nnz[m] = nnz[m];
    }
    void* base = nullptr;
    ZL_HIP(ctx, hipMalloc(&base, bytes ? bytes : 256));
    unsigned char* d = reinterpret_cast<unsigned char*>(base);
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    for (int m = 0; m < 3 && e == hipSuccess; m++) {
        e = hipMemcpyAsync(d + out->off_ptr[m], cs->row_ptr[m], (size_t)(nc + 1) * 4, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nnz[m]) {
            e = hipMemcpyAsync(d + out->off_col[m], cs->col[m], nnz[m] * 4, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipMemcpyAsync(d + out->off_val[m], cs->val[m], nnz[m] * 32, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL((k_fr_to_mont<FrP>), dim3((uint32_t)((nnz[m] + 255) / 256)), dim3(256), 0, st, (Fr*)(d + out->off_val[m]), (uint32_t)nnz[m]);
                e = hipGetLastError();
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { ctx->last_hip = (int)e; (void)hipFree(base); return ZL_EHIP; }
    out->d_base = base;
    out->n_constraints = nc;
    out->n_instance = cs->n_instance;
    out->n_witness = cs->n_witness;
    return ZL_OK;
}

// ---- per-key host state (round 6): the proof's fixed-base products and the folded C query ---------------------------------------------------------------
// A proof multiplies four points OF THE KEY by its blinding scalars (r delta1, s delta1, r s delta1, s delta2; the folded form below also s alpha1 and
// r beta1).  As variable-base products they were 256 doublings + 64 additions each (~0.3 ms in G1, ~0.7 ms in G2, one host thread per product); with the
// multiples d 2^(4 w) P, d = 1 .. 15, w = 0 .. 63, of a key point P tabulated once per key (960 affine points, batch-normalised: one inversion), a product is
// at most 64 mixed additions and no doubling: ~30 us in G1, ~100 us in G2.
template <class F>
struct FixedBase {
    std::vector<Affine<F>> tab;  // [64][15]
    XYZZ<F> p = XYZZ<F>::inf();
    bool tabulated = false;      // false: P is the point at infinity, or some small multiple of it is (a point outside the prime-order group): the plain product
    void build(const XYZZ<F>& point) {
        p = point;
        tabulated = false;
        tab.clear();
        if (p.is_inf()) return;
        std::vector<XYZZ<F>> t((size_t)64 * 15);
        XYZZ<F> base = p;
        for (int w = 0; w < 64; w++) {
            XYZZ<F>* row = &t[(size_t)w * 15];
            row[0] = base;
            for (int d = 1; d < 15; d++) {
                row[d] = row[d - 1];
                zl::add_full(row[d], base);
            }
            zl::add_full(base, row[14]);  // 16 base
        }
        for (const XYZZ<F>& e : t) if (e.is_inf()) return;
        // x = X / zz, y = Y / zzz with 1 / zz = (zz / zzz)^2 (zz^3 = zzz^2): one inversion for all the zzz (prefix products)
        const size_t n = t.size();
        std::vector<F> pre(n);
        pre[0] = t[0].zzz;
        for (size_t i = 1; i < n; i++) pre[i] = zl::mul(pre[i - 1], t[i].zzz);
        F run = zl::inv(pre[n - 1]);
        tab.resize(n);
        for (size_t i = n; i-- > 0;) {
            const F izzz = i ? zl::mul(run, pre[i - 1]) : run;
            run = zl::mul(run, t[i].zzz);
            const F iz = zl::mul(t[i].zz, izzz);
            tab[i] = Affine<F>{zl::canon(zl::mul(t[i].x, zl::sqr(iz))), zl::canon(zl::mul(t[i].y, izzz))};
        }
        tabulated = true;
    }
    XYZZ<F> mul(const uint32_t* k) const {  // k: 256 bits, little-endian words
        if (!tabulated) return zl::mul_scalar_w4(p, k);
        XYZZ<F> acc = XYZZ<F>::inf();
        for (int w = 0; w < 64; w++) {
            const uint32_t d = (k[w >> 3] >> ((w & 7) * 4)) & 15u;
            if (d) {
                const Affine<F>& e = tab[(size_t)w * 15 + d - 1];
                zl::add_mixed(acc, e.x, e.y, false);
            }
        }
        return acc;
    }
};
// The folded C query.  C = sum_w z_i L_i + sum h_i H_i + s A + r B1 - r s delta1 with A = alpha1 + sum z_i a_i + r delta1, B1 = beta1 + sum z_i b_i + s delta1, i.e.
//   C = [ sum_w z_i L_i + sum_i (s z_i) a_i + sum_i (r z_i) b_i + sum h_i H_i ]  +  s alpha1 + r beta1 + r s delta1:
// ONE multi-scalar multiplication over the concatenation l | a | b1 | h of the key's queries with the scalars z_w | s z | r z | h (the two scaled copies of z cost one
// Fr product per variable on the device), plus three fixed-base products.  B1 is never formed, the two VARIABLE-base products s A + r B1 (~0.3 ms of host time behind the
// a and b1 MSMs, on the critical path of a small proof) disappear, and a proof issues two G1 pipelines' worth of launches instead of four.  The price is n_variables more
// points through the bucket method (the a query is used twice) and scaled scalars that have lost the witness's zeros and ones -- so only proofs whose MSMs
// are latency, not throughput, take this form (ZL_TUNE_G16_FOLD_LOG_N).  The result is the same group element: the proof bytes do not change (tests/test_groth16.py).
#define ZL_G16_FOLD_LOG_N 14  // domains up to 2^14 fold on both curves (profiles/r06_fold_ab.log: 2^15 is even on BLS12-381 and a loss on BN254, 2^18 a loss of 14 %)
template <class G1, class G2>
struct G16KeyCache {
    using F1 = typename G1::F;
    using F2 = typename G2::F;
    std::vector<uint32_t> key_words;  // alpha1 | beta1 | delta1 | delta2 as handed in: the tables below belong to exactly these
    FixedBase<F1> alpha1, beta1, delta1;
    FixedBase<F2> delta2;
    zl_bases fold;  // l | a | b1 | h
    uint64_t fold_handles[4] = {0, 0, 0, 0};
    size_t fold_nv = 0, fold_nw = 0, fold_nh = 0;
    bool fold_built = false;
    void drop_fold() {
        if (fold.d_pts) (void)hipFree(fold.d_pts);
        if (fold.d_endo) (void)hipFree(fold.d_endo);
        if (fold.d_inf) (void)hipFree(fold.d_inf);
        fold = zl_bases{};
        fold_built = false;
    }
    ~G16KeyCache() { drop_fold(); }
};
template <class G1, class G2>
static std::vector<uint32_t> g16_key_words(const zl_g16_pk* pk) {
    constexpr size_t W1 = 2 * FieldIO<typename G1::F>::WORDS, W2 = 2 * FieldIO<typename G2::F>::WORDS;
    std::vector<uint32_t> w(3 * W1 + W2);
    memcpy(&w[0], pk->alpha_g1, W1 * 4);
    memcpy(&w[W1], pk->beta_g1, W1 * 4);
    memcpy(&w[2 * W1], pk->delta_g1, W1 * 4);
    memcpy(&w[3 * W1], pk->delta_g2, W2 * 4);
    return w;
}
// the cache of this key (built on first use), with the folded query for (nv, nw, nh) if `want_fold`.  bs: a, b1, h, l handles of the key
template <class G1, class G2>
static int g16_key_cache(zl_ctx* ctx, const zl_g16_pk* pk, const zl_bases* const* bs, bool want_fold, size_t nv, size_t nw, size_t nh,
                         std::shared_ptr<G16KeyCache<G1, G2>>* out) {
    using KC = G16KeyCache<G1, G2>;
    const std::vector<uint32_t> words = g16_key_words<G1, G2>(pk);
    const uint64_t hs[4] = {pk->l_query, pk->a_query, pk->b_g1_query, pk->h_query};
    std::lock_guard<std::mutex> lk(zl_bases_cache_mutex());  // (two lanes may prove over one key for the first time together)
    std::shared_ptr<KC> kc = std::static_pointer_cast<KC>(bs[3]->g16_cache);
    if (!kc || kc->key_words != words) {
        kc = std::make_shared<KC>();
        kc->key_words = words;
        kc->alpha1.build(affine_from_canon<G1>(pk->alpha_g1));
        kc->beta1.build(affine_from_canon<G1>(pk->beta_g1));
        kc->delta1.build(affine_from_canon<G1>(pk->delta_g1));
        kc->delta2.build(affine_from_canon<G2>(pk->delta_g2));
        bs[3]->g16_cache = kc;
    }
    if (want_fold && !(kc->fold_built && kc->fold_nv == nv && kc->fold_nw == nw && kc->fold_nh == nh && !memcmp(kc->fold_handles, hs, sizeof hs))) {
        if (kc.use_count() > 2) {  // a proof on another lane still works with the old folded query: a fresh cache object takes its place
            auto fresh = std::make_shared<KC>();
            fresh->key_words = kc->key_words;
            fresh->alpha1 = kc->alpha1;
            fresh->beta1 = kc->beta1;
            fresh->delta1 = kc->delta1;
            fresh->delta2 = kc->delta2;
            kc = fresh;
            bs[3]->g16_cache = kc;
        }
        kc->drop_fold();
        const zl_bases* parts[4] = {bs[3], bs[0], bs[1], bs[2]};
        const size_t first[4] = {0, 0, 0, 0}, n[4] = {nw, nv, nv, nh};
        zl_bases f;
        const int rc = ZL_DISPATCH(pk->curve, ZL_G1, zl_bases_concat, ctx, parts, first, n, 4, &f);
        if (rc) return rc;
        f.curve = (int)pk->curve;
        f.group = ZL_G1;
        kc->fold = f;
        memcpy(kc->fold_handles, hs, sizeof hs);
        kc->fold_nv = nv;
        kc->fold_nw = nw;
        kc->fold_nh = nh;
        kc->fold_built = true;
    }
    *out = kc;
    return ZL_OK;
}
// the scalars of the folded C query: out_zw = z[ni ..) (canonical), out_sz = s z, out_rz = r z (zm: z in Montgomery form, s and r canonical: the Montgomery product of
// z R and s is z s, canonical)
template <class FrP>
__global__ void __launch_bounds__(256) k_g16_fold_scalars(const Fp<FrP>* __restrict__ zc, const Fp<FrP>* __restrict__ zm, Fp<FrP> s, Fp<FrP> r, uint32_t ni, uint32_t nv,
                                                          Fp<FrP>* __restrict__ out_zw, Fp<FrP>* __restrict__ out_sz, Fp<FrP>* __restrict__ out_rz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const Fp<FrP> m = zm[i];
    out_sz[i] = i ? zl::mul(m, s) : s;  // (variable 0 is the constant ONE: its terms a_0, b_0 enter A and B1 unscaled, whatever the caller wrote there)
    out_rz[i] = i ? zl::mul(m, r) : r;
    if (i >= ni) out_zw[i - ni] = zc[i];
}

template <class G1, class G2>
// `witness` (optional): the assignment arrives in two pieces, `assignment` = the instance block and `witness` = the witness block, as the
// compiler holds them (openzl::Groth16::prove): two copies to the device instead of a host-side concatenation of tens of megabytes per proof
// wm_only: the assignment goes to the device and the witness map runs (z canonical at ctx->g16_z, h at ctx->g16_h), no MSM, no proof: the first step of a proof
// whose MSMs run on several devices (groth16_prove_sharded_t below); pk's handles are not looked at
static int groth16_prove_t(zl_ctx* ctx, const zl_g16_pk* pk, const zl_r1cs_dev* cs, const uint64_t* assignment, unsigned flags, const uint64_t* r,
                           const uint64_t* s, zl_g16_proof* out, const uint64_t* witness = nullptr, bool wm_only = false) {
    using FrP = typename G1::FrP;
    using Fr = Fp<FrP>;
    using F1 = typename G1::F;
    using F2 = typename G2::F;
    const uint32_t nc = cs->n_constraints, ni = cs->n_instance, nw = cs->n_witness;
    const uint64_t nv64 = (uint64_t)ni + nw;
    if (ni < 1 || nv64 >= (1ull << 31)) return ZL_EINVAL;
    const uint32_t nv = (uint32_t)nv64;
    unsigned log_n = 1;
    while ((1ull << log_n) < (uint64_t)nc + ni) log_n++;
    if (log_n > (unsigned)FrP::TWO_ADICITY || log_n > 28) return ZL_EINVAL;
    const uint32_t N = 1u << log_n;
    // handles
    const uint64_t hs[5] = {pk->a_query, pk->b_g1_query, pk->h_query, pk->l_query, pk->b_g2_query};
    const zl_bases* bs[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 5 && !wm_only; i++) {
        const zl_bases* bp = zl_find_bases(ctx, hs[i]);  // (a fork reads its parent's key)
        if (!bp) return ZL_EHANDLE;
        if (bp->curve != (int)pk->curve || bp->group != (i == 4 ? ZL_G2 : ZL_G1)) return ZL_EHANDLE;
        bs[i] = bp;
    }
    if (!wm_only && (bs[0]->n < nv || bs[1]->n < nv || bs[4]->n < nv || bs[2]->n < (size_t)N - 1 || bs[3]->n < nw)) return ZL_EINVAL;

    hipStream_t st = ctx->stream;
    static const bool trace = getenv("ZL_HOST_TRACE") != nullptr;  // developer aid: host-side phase times on stderr
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap_us = [&](const char* what) {
        if (trace) fprintf(stderr, "[zl_groth16 nc=%u] %-28s at %8.1f us\n", (unsigned)nc, what,
                           (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tp0).count() / 1e3);
    };
    ctx->g16_h = nullptr;  // the quotient of an earlier proof may live in a scratch block this call re-allocates
    ctx->g16_h_n = 0;
    if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    // ---- assignment + work vectors on the device (the matrices are already resident, zl_r1cs_upload) ------------------
    size_t bytes = 0;
    auto take = [&](size_t b) { size_t o = bytes; bytes += (b + 255) / 256 * 256; return o; };
    const size_t off_zc = take((size_t)nv * 32), off_zm = take((size_t)nv * 32);
    // a small proof's C comes from ONE folded query (G16KeyCache above): its scalars z_w | s z | r z | h are one vector, h (N entries written, N - 1 used) at its end
    const bool fold = !wm_only && log_n <= (unsigned)zl_tune("ZL_TUNE_G16_FOLD_LOG_N", ZL_G16_FOLD_LOG_N);
    const size_t fold_head = fold ? (size_t)nw + 2 * (size_t)nv : 0;
    const size_t off_a = take((size_t)N * 32), off_b = take((size_t)N * 32), off_c = take((size_t)N * 32), off_h = take((fold_head + N) * 32) + fold_head * 32;
    void* base;
    int rc;
    if ((rc = zl_scratch_get(ctx, 8, bytes, &base))) return rc;  // slots 0-7 belong to the MSM / NTT / staging paths
    unsigned char* d = reinterpret_cast<unsigned char*>(base);
    const unsigned char* dm = reinterpret_cast<const unsigned char*>(cs->d_base);
    const size_t* off_ptr = cs->off_ptr;
    const size_t* off_col = cs->off_col;
    const size_t* off_val = cs->off_val;
    Fr* d_zc = (Fr*)(d + off_zc);
    Fr* d_zm = (Fr*)(d + off_zm);
    auto h2d = [&](Fr* dst) -> hipError_t {
        if (!witness) return hipMemcpyAsync(dst, assignment, (size_t)nv * 32, hipMemcpyHostToDevice, st);
        hipError_t e = hipMemcpyAsync(dst, assignment, (size_t)ni * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && nw) e = hipMemcpyAsync(dst + ni, witness, (size_t)nw * 32, hipMemcpyHostToDevice, st);
        return e;
    };
    if (flags & ZL_MONT) {  // arkworks' in-memory assignment: Montgomery limbs; the canonical copy (MSM scalars) is made on the device
        ZL_HIP(ctx, h2d(d_zm));
        hipLaunchKernelGGL((k_fr_from_mont<FrP>), dim3((nv + 255) / 256), dim3(256), 0, st, d_zm, d_zc, nv);
    } else {
        ZL_HIP(ctx, h2d(d_zc));
        ZL_HIP(ctx, hipMemcpyAsync(d_zm, d_zc, (size_t)nv * 32, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL((k_fr_to_mont<FrP>), dim3((nv + 255) / 256), dim3(256), 0, st, d_zm, nv);
    }
    Fr *d_a = (Fr*)(d + off_a), *d_b = (Fr*)(d + off_b), *d_c = (Fr*)(d + off_c), *d_h = (Fr*)(d + off_h);
    Fr* d_fold = d_h - fold_head;
    if (fold) {
        Fr sf, rf;
        memcpy(sf.l, s, 32);
        memcpy(rf.l, r, 32);
        hipLaunchKernelGGL((k_g16_fold_scalars<FrP>), dim3((nv + 255) / 256), dim3(256), 0, st, d_zc, d_zm, sf, rf, ni, nv, d_fold, d_fold + nw, d_fold + nw + nv);
    }
    Fr* dv[3] = {d_a, d_b, d_c};
    // ---- witness map on its own context / stream: spmv, 3 x (ifft, coset fft), pointwise, coset ifft -> h -------------------------
    // It only needs z, like four of the five MSMs, so it runs beside them (memory- and latency-bound kernels in the shadow of the
    // bucket accumulation); the h MSM waits for ev_h inside the pipeline.
    { const int rc_aux = zl_ctx_aux_init(ctx); if (rc_aux) return rc_aux; }
    zl_ctx* wm = ctx->aux2;
    wm->ntt_fit_beside = wm_only ? 0 : 1;  // the witness map of a whole proof runs beside the G2 accumulation; on its own (sharded proofs) it takes the faster passes
    hipStream_t s_wm = wm->stream;
    hipEvent_t ev_z = wm->ev[0], ev_h = wm->ev[1];
    ZL_HIP(ctx, hipEventRecord(ev_z, st));
    const int timing_saved = ctx->timing_on;
    ctx->timing_on = 0;  // inner calls must not sync / overwrite the prover's events
    wm->timing_on = 0;
    // The ~16 launches of the witness map are issued by a persistent thread of the ctx while this one goes on to the MSMs (0.15 ms of host
    // time that stood in front of every MSM of a small proof).  h_recorded: 4 once ev_h is recorded (the h job, index 3 of the pipeline below,
    // may then wait for it), negative on failure.
    std::atomic<int> h_recorded{0};
    int rc_wm = ZL_OK;
    zl_worker& w_wm = zl_ctx_worker(ctx, 0);
    w_wm.run([&]() {
        auto fail = [&](int code) { rc_wm = code; h_recorded.store(-1, std::memory_order_release); };
        if (hipSetDevice(ctx->device) != hipSuccess || hipStreamWaitEvent(s_wm, ev_z, 0) != hipSuccess) return fail(ZL_EHIP);
        for (int m = 0; m < 3; m++) {
            // rows of >= 4 terms on average (A and B of the Poseidon circuit: ~13): eight lanes per row; C (one term per row): one lane
            if (cs->nnz[m] >= (size_t)4 * nc && zl_tune("ZL_TUNE_SPMV8", 1))
                hipLaunchKernelGGL((k_r1cs_spmv8<FrP>), dim3((uint32_t)(((uint64_t)N * 8 + 255) / 256)), dim3(256), 0, s_wm, (const uint32_t*)(dm + off_ptr[m]), (const uint32_t*)(dm + off_col[m]),
                                   (const Fr*)(dm + off_val[m]), d_zm, nc, m == 0 ? ni : 0u, N, dv[m]);
            else
                hipLaunchKernelGGL((k_r1cs_spmv<FrP>), dim3((N + 255) / 256), dim3(256), 0, s_wm, (const uint32_t*)(dm + off_ptr[m]), (const uint32_t*)(dm + off_col[m]),
                                   (const Fr*)(dm + off_val[m]), d_zm, nc, m == 0 ? ni : 0u, N, dv[m]);
        }
        if (hipGetLastError() != hipSuccess) return fail(ZL_EHIP);
        int r;
        // a, b, c go through their inverse transform and their coset transform three at a time (one launch per pass, grid.y = vector): a transform of <= 2^16 elements
        // is a chain of passes of ~20 us each whatever its size (16 workgroups at 2^14), so six of the witness map's seven transforms cost the latency of two
        // (interleaved A/B, profiles/r06_ntt_batch_ab.log: 1 882 constraints 1.00 -> 0.86 ms, 14 977: 1.58 -> 1.45 (BN254 1.25 -> 1.10), 2^16 - 2^18: -2 %; at 2^20 the passes fill
        // the machine and three at a time is 1-2 % SLOWER beside the G2 accumulation: domains up to 2^18 only)
        if (off_c - off_b == off_b - off_a && log_n <= (unsigned)zl_tune("ZL_TUNE_NTT_BATCH_LOG_N", 18)) {
            if ((r = zl_ntt_run_batch(wm, pk->curve, d_a, log_n, ZL_MONT | ZL_INVERSE, 3, off_b - off_a))) return fail(r);
            if ((r = zl_ntt_run_batch(wm, pk->curve, d_a, log_n, ZL_MONT | ZL_COSET, 3, off_b - off_a))) return fail(r);
        } else {
            for (int m = 0; m < 3; m++) {
                if ((r = zl_ntt_run(wm, pk->curve, dv[m], log_n, ZL_MONT | ZL_INVERSE))) return fail(r);
                if ((r = zl_ntt_run(wm, pk->curve, dv[m], log_n, ZL_MONT | ZL_COSET))) return fail(r);
            }
        }
        Fr g;
        for (int i = 0; i < Fr::N; i++) g.l[i] = FrP::generator(i);
        Fr gN = g;
        for (unsigned i = 0; i < log_n; i++) gN = zl::sqr(gN);
        const Fr zinv = zl::inv(zl::sub(gN, Fr::one()));
        hipLaunchKernelGGL((k_qap_pointwise<FrP>), dim3((N + 255) / 256), dim3(256), 0, s_wm, d_a, d_b, d_c, zinv, N);
        if ((r = zl_ntt_run(wm, pk->curve, d_a, log_n, ZL_MONT | ZL_INVERSE | ZL_COSET))) return fail(r);
        hipLaunchKernelGGL((k_fr_from_mont<FrP>), dim3((N + 255) / 256), dim3(256), 0, s_wm, d_a, d_h, N);
        if (hipGetLastError() != hipSuccess || hipEventRecord(ev_h, s_wm) != hipSuccess) return fail(ZL_EHIP);
        h_recorded.store(4, std::memory_order_release);
    });
    // (every return below first waits for that thread: it works on this frame)
    auto wm_fail = [&](int code) { w_wm.wait(); (void)hipStreamSynchronize(s_wm); ctx->timing_on = timing_saved; return code; };
    if (wm_only) {
        w_wm.wait();
        const hipError_t e1 = hipStreamSynchronize(s_wm), e2 = hipStreamSynchronize(st);
        ctx->timing_on = timing_saved;
        if (rc_wm) return rc_wm;
        if (e1 != hipSuccess || e2 != hipSuccess) { ctx->last_hip = (int)(e1 != hipSuccess ? e1 : e2); return ZL_EHIP; }
        ctx->g16_h = d_h;
        ctx->g16_h_n = N;
        ctx->g16_z = d_zc;
        return ZL_OK;
    }
    // ---- the five MSMs ----------------------------------------------------------------------------------------------
    uint64_t part[5][ZL_PARTIAL_WORDS];
    const unsigned char* zc = reinterpret_cast<const unsigned char*>(d_zc);
    // The G2 MSM runs on an auxiliary stream with its own scratch while the G1 MSMs run on the main streams: the two fill each other's gaps (54 -> ~35 ms at 2^20).
    // Nobody waits for z on the host: the G1 pipeline's streams all start behind an event it records on `st` (zl_msm.hip: ev_begin), the witness map and the G2
    // stream wait for ev_z (a small proof spent ~35 us here, in front of every launch of its MSMs)
    // first points of the a / b queries (index 0 pairs with z[0] = 1); fetched here, before the MSM streams are busy (the download uses the
    // ctx's own stream and sort scratch)
    uint64_t a0_xy[12], b0_xy[12], b20_xy[24];
    {
        // static per key: fetched from the device on the first proof only
        const struct { const zl_bases* b; int group; uint64_t* out; size_t words; } firsts[3] = {
            {bs[0], ZL_G1, a0_xy, 12}, {bs[1], ZL_G1, b0_xy, 12}, {bs[4], ZL_G2, b20_xy, 24}};
        std::lock_guard<std::mutex> cache_lk(zl_bases_cache_mutex());  // (two lanes may prove over one key for the first time together)
        for (const auto& f : firsts) {
            if (f.b->first_xy.empty()) {
                uint64_t tmp[24] = {0};
                if ((rc = ZL_DISPATCH(pk->curve, f.group, zl_bases_download, ctx, *f.b, 0, 1, tmp))) return wm_fail(rc);
                f.b->first_xy.assign(tmp, tmp + f.words);
            }
            memcpy(f.out, f.b->first_xy.data(), f.words * 8);
        }
    }
    // the key's host tables (and, for a small proof, its folded C query): built by the first proof over the key
    std::shared_ptr<G16KeyCache<G1, G2>> kc;
    if ((rc = g16_key_cache<G1, G2>(ctx, pk, bs, fold, nv, nw, (size_t)N - 1, &kc))) return wm_fail(rc);
    // host work that does not depend on the MSMs -- the fixed-base products r delta1, s delta2 and (folded form) s alpha1 + r beta1 + r s delta1, or (four-MSM form)
    // s delta1 and -r s delta1 -- runs on two threads while the device is busy: at most 64 mixed additions each from the key's tables
    uint32_t rw[8], sw[8];
    memcpy(rw, r, 32);
    memcpy(sw, s, 32);
    XYZZ<F1> r_delta1, s_delta1, rs_delta, c_fixed;
    XYZZ<F2> s_delta2;
    // (r s) delta1 from the product r s in Fr, so that the products are independent
    uint32_t rsw[8];
    {
        using FrF = Fp<typename G1::FrP>;
        FrF rm, sm;
        memcpy(rm.l, rw, 32);
        memcpy(sm.l, sw, 32);
        const FrF rs = zl::from_mont(zl::mul(zl::to_mont(rm), zl::to_mont(sm)));
        memcpy(rsw, rs.l, 32);
    }
    std::thread pre([&]() {
        r_delta1 = kc->delta1.mul(rw);
        lap_us("r delta1 done");
        rs_delta = kc->delta1.mul(rsw);
        if (fold) {
            c_fixed = kc->alpha1.mul(sw);
            zl::add_full(c_fixed, kc->beta1.mul(rw));
            zl::add_full(c_fixed, rs_delta);
        } else {
            zl::neg_inplace(rs_delta);
            s_delta1 = kc->delta1.mul(sw);
        }
        lap_us("G1 fixed-base products done");
    });
    std::thread pre_g2([&]() { s_delta2 = kc->delta2.mul(sw); lap_us("s delta2 done"); });
    lap_us("z on device, host pre started");
    int rc_g2 = ZL_OK;
    XYZZ<F1> g_a = XYZZ<F1>::inf(), g1_b = XYZZ<F1>::inf(), g_c = XYZZ<F1>::inf();
    bool have_c = false;
    uint64_t a_words[12] = {0}, b_words[24] = {0};
    uint8_t a_inf = 0, b_inf = 0;
    {
        zl_ctx* aux = ctx->aux;
        const zl_bases* b2 = bs[4];
        const int curve = pk->curve;
        uint64_t* out2 = part[4];
        zl_worker& g2 = zl_ctx_worker(ctx, 1);
        // (Tried: every MSM of a folded proof behind the witness map, so that the accumulations do not sit on the SIMDs its short passes need -- slower at every size, 235 constraints
        // 0.67 -> 0.74, 14 977: 1.58 -> 1.61 (profiles/r06_gate_ab.log): a 2^14 transform is two passes of ~20 us on 16 workgroups whether or not the machine is busy.  What helped
        // is running a, b, c through each transform in one launch, above.)
        g2.run([&, aux, b2, curve, out2]() {
            if (hipSetDevice(aux->device) != hipSuccess || hipStreamWaitEvent(aux->stream, ev_z, 0) != hipSuccess) { rc_g2 = ZL_EHIP; return; }
            rc_g2 = ZL_DISPATCH(curve, ZL_G2, zl_msm_run, aux, *b2, 1, zc + 32, nv - 1, out2);
            if (rc_g2 != ZL_OK) return;
            // B = beta2 + sum z_i b2_i + s delta2 is complete here: assembled and normalised (the Fq2 inversion) on this thread, beside the G1 pipeline
            pre_g2.join();
            XYZZ<F2> g2_b = s_delta2;
            zl::add_full(g2_b, affine_from_canon<G2>(b20_xy));
            zl::add_full(g2_b, from_partial<F2>(out2));
            zl::add_full(g2_b, affine_from_canon<G2>(pk->beta_g2));
            store_canon<G2>(b_words, &b_inf, g2_b);
            lap_us("B done");
        });
        if (fold) {
            // folded form: the a MSM (for A) and ONE MSM over l | a | b1 | h for all of C's sums; the second waits for the witness map's event
            const zl_bases* jb[2] = {bs[0], &kc->fold};
            const size_t jf[2] = {1, 0};
            const void* js[2] = {zc + 32, d_fold};
            const size_t jn[2] = {(size_t)nv - 1, fold_head + (size_t)N - 1};
            const hipEvent_t jw[2] = {nullptr, ev_h};
            uint64_t jp[2][ZL_PARTIAL_WORDS];
            const std::function<void(size_t)> on_done = [&](size_t i) {
                if (i != 0) return;
                pre.join();
                g_a = r_delta1;
                zl::add_full(g_a, affine_from_canon<G1>(a0_xy));
                zl::add_full(g_a, from_partial<F1>(jp[0]));
                zl::add_full(g_a, affine_from_canon<G1>(pk->alpha_g1));
                store_canon<G1>(a_words, &a_inf, g_a);  // A is final: its normalisation (one inversion) runs under the second MSM
                g_c = c_fixed;
                have_c = true;
                lap_us("A done");
            };
            rc = ZL_DISPATCH(pk->curve, ZL_G1, zl_msm_run_jobs, ctx, jb, jf, js, jn, jw, 2, &jp[0][0], (const std::atomic<int>*)&h_recorded, &on_done);
            memcpy(part[0], jp[0], sizeof jp[0]);
            memcpy(part[3], jp[1], sizeof jp[1]);                 // every sum of C
            { const XYZZ<F1> none = XYZZ<F1>::inf(); memset(part[2], 0, sizeof part[2]); memcpy(part[2], &none, sizeof none); }
        } else {
        // the four G1 MSMs as one pipeline (sort | accumulate | tail of consecutive MSMs overlap, zl_msm.hip): l, a, b1 need only z;
        // the h job waits for the witness map's event
        {
            // order a, b1, l, h: A and B1 are complete after the first two jobs (their product s A + r B1 then runs under the other two)
            const zl_bases* jb[4] = {bs[0], bs[1], bs[3], bs[2]};
            const size_t jf[4] = {1, 1, 0, 0};
            const void* js[4] = {zc + 32, zc + 32, zc + (size_t)ni * 32, d_h};
            const size_t jn[4] = {(size_t)nv - 1, (size_t)nv - 1, (size_t)nw, (size_t)N - 1};
            const hipEvent_t jw[4] = {nullptr, nullptr, nullptr, ev_h};
            uint64_t jp[4][ZL_PARTIAL_WORDS];
            // A and B1 are complete once the a and b1 MSMs (jobs 0, 1) are: their products s A and r B1 (256 doublings + 64 additions each on the host) run on
            // the pipeline's completion thread and one more while the h MSM and the G2 MSM are still on the device
            const std::function<void(size_t)> on_done = [&](size_t i) {
                if (i != 1) return;
                lap_us("a, b1 delivered");
                pre.join();
                g_a = r_delta1;
                zl::add_full(g_a, affine_from_canon<G1>(a0_xy));
                zl::add_full(g_a, from_partial<F1>(jp[0]));
                zl::add_full(g_a, affine_from_canon<G1>(pk->alpha_g1));
                g1_b = s_delta1;
                zl::add_full(g1_b, affine_from_canon<G1>(b0_xy));
                zl::add_full(g1_b, from_partial<F1>(jp[1]));
                zl::add_full(g1_b, affine_from_canon<G1>(pk->beta_g1));
                lap_us("a, b1 in: s A + r B1 starts");
                {   // s A + r B1 as two single-scalar products side by side (0.3 ms) instead of one interleaved double-scalar product (0.45 ms)
                    XYZZ<F1> rb = XYZZ<F1>::inf();
                    std::thread t_rb([&]() { rb = zl::mul_scalar_w4(g1_b, rw); });
                    g_c = zl::mul_scalar_w4(g_a, sw);
                    t_rb.join();
                    zl::add_full(g_c, rb);
                }
                zl::add_full(g_c, rs_delta);
                store_canon<G1>(a_words, &a_inf, g_a);  // A is final: its normalisation (one inversion) leaves the critical path too
                have_c = true;
                lap_us("s A + r B1 done");
            };
            rc = ZL_DISPATCH(pk->curve, ZL_G1, zl_msm_run_jobs, ctx, jb, jf, js, jn, jw, 4, &jp[0][0], (const std::atomic<int>*)&h_recorded, &on_done);
            memcpy(part[0], jp[0], sizeof jp[0]);
            memcpy(part[1], jp[1], sizeof jp[1]);
            memcpy(part[3], jp[2], sizeof jp[2]);
            memcpy(part[2], jp[3], sizeof jp[3]);
        }
        }
        lap_us("G1 pipeline returned");
        w_wm.wait();
        (void)hipStreamSynchronize(s_wm);  // also on the error path: nothing of this proof may still be running
        g2.wait();
        if (!rc) rc = rc_wm;
        lap_us("G2 joined");
    }
    if (pre.joinable()) pre.join();
    if (pre_g2.joinable()) pre_g2.join();
    if (!rc) rc = rc_g2;
    if (!rc && !have_c) rc = ZL_EHIP;  // (the completion callback did not run: cannot happen after a successful pipeline)
    ctx->timing_on = timing_saved;
    if (rc) return rc;
    if (ctx->timing_on) {
        ZL_HIP(ctx, hipEventRecord(ctx->ev[1], st));
        ZL_HIP(ctx, hipStreamSynchronize(st));
        ctx->timing = zl_timing{};
        ZL_HIP(ctx, hipEventElapsedTime(&ctx->timing.total_ms, ctx->ev[0], ctx->ev[1]));
        ctx->timing.launches = 5;
    }
    ctx->g16_h = d_h;
    ctx->g16_h_n = N;
    // ---- host assembly: A (and, in the four-MSM form, s A + r B1 - r s delta1) were formed by the completion callback above and B by the G2 worker; C needs the last MSMs
    zl::add_full(g_c, from_partial<F1>(part[3]));
    zl::add_full(g_c, from_partial<F1>(part[2]));
    lap_us("assembly");
    memset(out, 0, sizeof *out);
    memcpy(out->a, a_words, sizeof a_words);
    out->a_inf = a_inf;
    memcpy(out->b, b_words, sizeof b_words);
    out->b_inf = b_inf;
    store_canon<G1>(out->c, &out->c_inf, g_c);
    lap_us("proof normalised");
    return ZL_OK;
}

// ---- one proof over the G devices of an mctx ---------------------------------------------------------------------------
// Rank g holds, on zl_mctx_ctx(m, g), bases handles with ITS contiguous slice of every query (zl_g16_shard); rank 0 also holds the constraint matrices.
//   1. rank 0: assignment -> device, witness map (spmv, 7 NTTs) -> h                                        (groth16_prove_t, wm_only)
//   2. every rank, on its own host thread: its slices of z and h arrive by device-to-device copy, then its five partial MSMs (a, b1, l, h in G1 and b2
//      in G2) through the single-device pipeline (zl_msm_run) -> five un-normalised partial sums per rank
//   3. host: the partials of each MSM are folded over the ranks (what an all-gather + fold does between processes: EC addition is not a collective's
//      reduce op) and the proof is assembled exactly as the single-device prover does
// Same proof, byte for byte, as zl_groth16_prove_resident with the same (r, s) (tests/test_gpu_multi.py).  Functional: the MSM phase of a proof shards like
// any MSM (512 B of partials per rank and MSM); the witness map stays on one device (the NTTs of one proof are replicas work, not sharded work).
template <class G1, class G2>
static int groth16_prove_sharded_t(zl_mctx* m, const zl_g16_pk* pk, const zl_g16_shard* shards, const zl_r1cs_dev* cs, const uint64_t* assignment, unsigned flags,
                                   const uint64_t* r, const uint64_t* s, zl_g16_proof* out) {
    using FrP = typename G1::FrP;
    using F1 = typename G1::F;
    using F2 = typename G2::F;
    const int G = zl_mctx_size(m);
    zl_ctx* ctx0 = zl_mctx_ctx(m, 0);
    const uint32_t ni = cs->n_instance, nw = cs->n_witness, nv = ni + nw;
    unsigned log_n = 1;
    while ((1ull << log_n) < (uint64_t)cs->n_constraints + ni) log_n++;
    const size_t N = (size_t)1 << log_n;
    // the slices must tile [0, nv), [0, nw) and [0, N - 1) in rank order
    size_t v = 0, w = 0, hq = 0;
    for (int g = 0; g < G; g++) {
        if (shards[g].var_first != v || shards[g].wit_first != w || shards[g].h_first != hq) return ZL_EINVAL;
        v += shards[g].var_count; w += shards[g].wit_count; hq += shards[g].h_count;
    }
    if (v != nv || w != nw || hq != N - 1) return ZL_EINVAL;
    int rc;
    ZL_HIP(ctx0, hipSetDevice(ctx0->device));
    if ((rc = groth16_prove_t<G1, G2>(ctx0, pk, cs, assignment, flags, r, s, nullptr, nullptr, true))) return rc;
    const unsigned char* d_z0 = reinterpret_cast<const unsigned char*>(ctx0->g16_z);
    const unsigned char* d_h0 = reinterpret_cast<const unsigned char*>(ctx0->g16_h);
    std::vector<std::array<uint64_t, 5 * ZL_PARTIAL_WORDS>> parts((size_t)G);
    std::vector<int> rcs((size_t)G, ZL_OK);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++)
        th.emplace_back([&, g]() {
            zl_ctx* c = zl_mctx_ctx(m, g);
            const zl_g16_shard& sh = shards[g];
            auto fail = [&](int code) { rcs[(size_t)g] = code; };
            if (hipSetDevice(c->device) != hipSuccess) return fail(ZL_EHIP);
            // this rank's scalars: [variables | witnesses | quotient coefficients], canonical, copied from rank 0's device
            void* d = nullptr;
            const size_t nvar = sh.var_count, nwit = sh.wit_count, nh = sh.h_count;
            int r2 = zl_scratch_get(c, 7, (nvar + nwit + nh + 1) * 32, &d);
            if (r2) return fail(r2);
            unsigned char* ds = reinterpret_cast<unsigned char*>(d);
            hipError_t e = hipSuccess;
            if (nvar) e = hipMemcpyPeerAsync(ds, c->device, d_z0 + sh.var_first * 32, ctx0->device, nvar * 32, c->stream);
            if (e == hipSuccess && nwit) e = hipMemcpyPeerAsync(ds + nvar * 32, c->device, d_z0 + ((size_t)ni + sh.wit_first) * 32, ctx0->device, nwit * 32, c->stream);
            if (e == hipSuccess && nh) e = hipMemcpyPeerAsync(ds + (nvar + nwit) * 32, c->device, d_h0 + sh.h_first * 32, ctx0->device, nh * 32, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { c->last_hip = (int)e; return fail(ZL_EHIP); }
            const struct { uint64_t handle; int group; const unsigned char* sc; size_t n; } jobs[5] = {
                {sh.a_query, ZL_G1, ds, nvar}, {sh.b_g1_query, ZL_G1, ds, nvar}, {sh.h_query, ZL_G1, ds + (nvar + nwit) * 32, nh},
                {sh.l_query, ZL_G1, ds + nvar * 32, nwit}, {sh.b_g2_query, ZL_G2, ds, nvar}};
            for (int j = 0; j < 5; j++) {
                uint64_t* outp = parts[(size_t)g].data() + (size_t)j * ZL_PARTIAL_WORDS;
                memset(outp, 0, ZL_PARTIAL_WORDS * 8);
                if (jobs[j].n == 0) {  // an empty slice contributes the point at infinity
                    if (jobs[j].group == ZL_G1) { const XYZZ<F1> z = XYZZ<F1>::inf(); memcpy(outp, &z, sizeof z); }
                    else { const XYZZ<F2> z = XYZZ<F2>::inf(); memcpy(outp, &z, sizeof z); }
                    continue;
                }
                auto it = c->bases.find(jobs[j].handle);
                if (it == c->bases.end() || it->second.curve != (int)pk->curve || it->second.group != jobs[j].group || it->second.n < jobs[j].n) return fail(ZL_EHANDLE);
                r2 = ZL_DISPATCH(pk->curve, jobs[j].group, zl_msm_run, c, it->second, 0, jobs[j].sc, jobs[j].n, outp);
                if (r2) return fail(r2);
            }
        });
    for (auto& t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g]) return rcs[(size_t)g];
    // fold over the ranks, then the assembly of create_proof_with_assignment
    XYZZ<F1> sum1[4];
    XYZZ<F2> sum2 = XYZZ<F2>::inf();
    for (auto& x : sum1) x = XYZZ<F1>::inf();
    for (int g = 0; g < G; g++) {
        for (int j = 0; j < 4; j++) zl::add_full(sum1[j], from_partial<F1>(parts[(size_t)g].data() + (size_t)j * ZL_PARTIAL_WORDS));
        zl::add_full(sum2, from_partial<F2>(parts[(size_t)g].data() + (size_t)4 * ZL_PARTIAL_WORDS));
    }
    uint32_t rw[8], sw[8], rsw[8];
    memcpy(rw, r, 32);
    memcpy(sw, s, 32);
    {
        using FrF = Fp<FrP>;
        FrF rm, sm;
        memcpy(rm.l, rw, 32);
        memcpy(sm.l, sw, 32);
        const FrF rs = zl::from_mont(zl::mul(zl::to_mont(rm), zl::to_mont(sm)));
        memcpy(rsw, rs.l, 32);
    }
    const XYZZ<F1> delta1 = affine_from_canon<G1>(pk->delta_g1);
    const XYZZ<F2> delta2 = affine_from_canon<G2>(pk->delta_g2);
    XYZZ<F1> g_a = zl::mul_scalar_w4(delta1, rw);           // A = alpha + sum z_i a_i + r delta
    zl::add_full(g_a, sum1[0]);
    zl::add_full(g_a, affine_from_canon<G1>(pk->alpha_g1));
    XYZZ<F1> g1_b = zl::mul_scalar_w4(delta1, sw);          // B1 = beta + sum z_i b_i + s delta
    zl::add_full(g1_b, sum1[1]);
    zl::add_full(g1_b, affine_from_canon<G1>(pk->beta_g1));
    XYZZ<F2> g2_b = zl::mul_scalar_w4(delta2, sw);          // B = the same in G2
    zl::add_full(g2_b, sum2);
    zl::add_full(g2_b, affine_from_canon<G2>(pk->beta_g2));
    XYZZ<F1> g_c = zl::mul_scalar_w4(g_a, sw);              // C = s A + r B1 - r s delta + sum_w z_w l_w + sum h_i H_i
    zl::add_full(g_c, zl::mul_scalar_w4(g1_b, rw));
    XYZZ<F1> rs_delta = zl::mul_scalar_w4(delta1, rsw);
    zl::neg_inplace(rs_delta);
    zl::add_full(g_c, rs_delta);
    zl::add_full(g_c, sum1[3]);
    zl::add_full(g_c, sum1[2]);
    memset(out, 0, sizeof *out);
    store_canon<G1>(out->a, &out->a_inf, g_a);
    store_canon<G2>(out->b, &out->b_inf, g2_b);
    store_canon<G1>(out->c, &out->c_inf, g_c);
    return ZL_OK;
}
extern "C" int zl_groth16_prove_sharded(zl_mctx* m, const zl_g16_pk* pk, const zl_g16_shard* shards, uint64_t r1cs_handle_rank0, const uint64_t* assignment,
                                        unsigned flags, const uint64_t* r, const uint64_t* s, zl_g16_proof* out) {
    if (!m || !pk || !shards || !assignment || !r || !s || !out || (flags & ~ZL_MONT)) return ZL_EINVAL;
    if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->beta_g2 || !pk->delta_g2) return ZL_EINVAL;
    zl_ctx* ctx0 = zl_mctx_ctx(m, 0);
    if (!ctx0) return ZL_EINVAL;
    auto it = ctx0->r1cs.find(r1cs_handle_rank0);
    if (it == ctx0->r1cs.end() || it->second.curve != (int)pk->curve) return ZL_EHANDLE;
    if (pk->curve == ZL_BLS12_381) return groth16_prove_sharded_t<BlsG1, BlsG2>(m, pk, shards, &it->second, assignment, flags, r, s, out);
    if (pk->curve == ZL_BN254) return groth16_prove_sharded_t<BnG1, BnG2>(m, pk, shards, &it->second, assignment, flags, r, s, out);
    return ZL_EINVAL;
}

// Full structural validation of a caller-supplied CSR, once per upload (O(nnz) on the host): row_ptr starts at 0 and is monotone,
// every column index names a variable.  k_r1cs_spmv then never reads val / z out of bounds, whatever the caller passed.
static bool r1cs_view_ok(const zl_r1cs* cs) {
    if (!cs || cs->n_instance < 1) return false;
    const uint64_t nv = (uint64_t)cs->n_instance + cs->n_witness;
    if (nv >= (1ull << 31)) return false;
    for (int m = 0; m < 3; m++) {
        const uint32_t* rp = cs->row_ptr[m];
        if (!rp || rp[0] != 0) return false;
        for (uint32_t i = 0; i < cs->n_constraints; i++)
            if (rp[i + 1] < rp[i]) return false;
        const uint32_t nnz = rp[cs->n_constraints];
        if (nnz && (!cs->col[m] || !cs->val[m])) return false;
        for (uint32_t k = 0; k < nnz; k++)
            if (cs->col[m][k] >= nv) return false;
    }
    return true;
}
extern "C" int zl_r1cs_upload(zl_ctx* ctx, zl_curve_t curve, const zl_r1cs* cs, uint64_t* handle_out) {
    if (!ctx || !handle_out || !r1cs_view_ok(cs)) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    zl_r1cs_dev dev{};
    dev.curve = curve;
    const int rc = curve == ZL_BLS12_381 ? r1cs_upload_t<BLS12_381_Fr>(ctx, cs, &dev) : r1cs_upload_t<BN254_Fr>(ctx, cs, &dev);
    if (rc) return rc;
    uint64_t h;
    {
        std::unique_lock<std::shared_mutex> lk(ctx->maps_mu);
        h = ctx->next_handle++;
        ctx->r1cs[h] = dev;
    }
    *handle_out = h;
    return ZL_OK;
}
extern "C" int zl_r1cs_free(zl_ctx* ctx, uint64_t handle) {
    if (!ctx) return ZL_EINVAL;
    auto it = ctx->r1cs.find(handle);
    if (it == ctx->r1cs.end()) return ZL_EHANDLE;
    zl_ctx_release_idle_lane(ctx);
    if (ctx->forks.load() > 0) return ZL_EINVAL;  // a fork may be reading it: destroy the forks first
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (it->second.d_base) (void)hipFree(it->second.d_base);
    ctx->r1cs.erase(it);
    return ZL_OK;
}
extern "C" int zl_groth16_prove_resident(zl_ctx* ctx, const zl_g16_pk* pk, uint64_t r1cs_handle, const uint64_t* assignment, unsigned flags,
                                         const uint64_t* r, const uint64_t* s, zl_g16_proof* out) {
    if (!ctx || !pk || !assignment || !r || !s || !out || (flags & ~ZL_MONT)) return ZL_EINVAL;
    if (!pk->alpha_g1 || !pk->beta_g1 || !pk->delta_g1 || !pk->beta_g2 || !pk->delta_g2) return ZL_EINVAL;
    const zl_r1cs_dev* rd = zl_find_r1cs(ctx, r1cs_handle);
    if (!rd || rd->curve != (int)pk->curve) return ZL_EHANDLE;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    if (pk->curve == ZL_BLS12_381) return groth16_prove_t<BlsG1, BlsG2>(ctx, pk, rd, assignment, flags, r, s, out);
    if (pk->curve == ZL_BN254) return groth16_prove_t<BnG1, BnG2>(ctx, pk, rd, assignment, flags, r, s, out);
    return ZL_EINVAL;
}
// internal (zl_host.hip): the assignment as the compiler holds it -- instance block and witness block, Montgomery limbs
int zl_groth16_prove_split(zl_ctx* ctx, const zl_g16_pk* pk, uint64_t r1cs_handle, const uint64_t* instance, const uint64_t* witness, const uint64_t* r,
                           const uint64_t* s, zl_g16_proof* out) {
    if (!ctx || !pk || !instance || !r || !s || !out) return ZL_EINVAL;
    const zl_r1cs_dev* rd = zl_find_r1cs(ctx, r1cs_handle);
    if (!rd || rd->curve != (int)pk->curve) return ZL_EHANDLE;
    if (rd->n_witness && !witness) return ZL_EINVAL;
    ZL_HIP(ctx, hipSetDevice(ctx->device));
    static const uint64_t none[4] = {0, 0, 0, 0};
    if (pk->curve == ZL_BLS12_381) return groth16_prove_t<BlsG1, BlsG2>(ctx, pk, rd, instance, ZL_MONT, r, s, out, witness ? witness : none);
    if (pk->curve == ZL_BN254) return groth16_prove_t<BnG1, BnG2>(ctx, pk, rd, instance, ZL_MONT, r, s, out, witness ? witness : none);
    return ZL_EINVAL;
}
extern "C" int zl_groth16_prove(zl_ctx* ctx, const zl_g16_pk* pk, const zl_r1cs* cs, const uint64_t* assignment, const uint64_t* r, const uint64_t* s,
                                zl_g16_proof* out) {
    if (!ctx || !pk || !r1cs_view_ok(cs)) return ZL_EINVAL;
    uint64_t h = 0;
    int rc = zl_r1cs_upload(ctx, pk->curve, cs, &h);
    if (rc) return rc;
    rc = zl_groth16_prove_resident(ctx, pk, h, assignment, ZL_CANON, r, s, out);
    (void)zl_r1cs_free(ctx, h);
    return rc;
}
extern "C" int zl_groth16_last_h(zl_ctx* ctx, uint64_t* out, size_t n) {
    if (!ctx || !out || !ctx->g16_h || n > ctx->g16_h_n) return ZL_EINVAL;
    ZL_HIP(ctx, hipMemcpyAsync(out, ctx->g16_h, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZL_OK;
}
