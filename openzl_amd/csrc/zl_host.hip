// zl_host.hip -- implementation of the host-side plugin mirror (zl_host.h) + its C-ABI hooks (include/zl_backend.h,
// "host mirror" section).  Host code only; compiled by hipcc because it shares zl_field.h with the kernels.
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <chrono>
#include <new>
#include <atomic>
#include <thread>
#include "zl_host.h"
#include "zl_serialize.h"

// zl_groth16.hip: the prover with the assignment in the compiler's two pieces (instance block, witness block; Montgomery limbs)
int zl_groth16_prove_split(zl_ctx* ctx, const zl_g16_pk* pk, uint64_t r1cs_handle, const uint64_t* instance, const uint64_t* witness, const uint64_t* r,
                           const uint64_t* s, zl_g16_proof* out);  // zl_groth16.hip

namespace openzl {

template <class FrP>
static void to_canon_words(uint64_t* out, const Fp<FrP>& mont) {
    const Fp<FrP> c = zl::from_mont(mont);
    memcpy(out, c.l, 32);
}

// fn(lo, hi, chunk) over [0, n) on up to 64 host threads (the setup path is O(constraints) field work: Lagrange coefficients, the
// transposed sparse products, exponent vectors; single-threaded it cost 3 s of a 4 s compile at 10^6 constraints)
template <class Fn>
static void parallel_chunks(size_t n, size_t min_chunk, Fn fn, size_t max_threads = 64) {
    if (const char* e = getenv("ZL_HOST_THREADS")) max_threads = std::max(1, atoi(e));
    size_t nt = std::min<size_t>(std::max<unsigned>(1u, std::thread::hardware_concurrency()), max_threads);
    nt = std::min(nt, std::max<size_t>(1, n / std::max<size_t>(1, min_chunk)));
    if (nt <= 1) { fn((size_t)0, n, (size_t)0); return; }
    const size_t per = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) {
        const size_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back([=]() { fn(lo, hi, t); });
    }
    for (auto& x : th) x.join();
}
template <class FrP>
static Fp<FrP> pow_u64(const Fp<FrP>& a, uint64_t e) {
    const uint32_t w[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
    return zl::pow_words(a, w, 64);
}

template <class FrP>
void R1CS<FrP>::replicate_rows(size_t r0, size_t r1, size_t copies, uint32_t w_from, uint32_t shift, const std::vector<F>& new_witness_values) {
    const size_t rows = r1 - r0, base = A_.size();
    witness_.insert(witness_.end(), new_witness_values.begin(), new_witness_values.end());
    A_.resize(base + rows * copies);
    B_.resize(base + rows * copies);
    C_.resize(base + rows * copies);
    parallel_chunks(rows * copies, 256, [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; i++) {
            const size_t j = i / rows + 1, r = r0 + i % rows;
            const uint32_t delta = (uint32_t)(j * shift);
            A_[base + i] = shifted(A_[r], w_from, delta);
            B_[base + i] = shifted(B_[r], w_from, delta);
            C_[base + i] = shifted(C_[r], w_from, delta);
        }
    }, 16);  // millions of small allocations: beyond ~16 threads the page-fault / allocator contention costs more than it buys (measured: 0.7 s at 16, 2.2 s at 64)
}

template <class FrP>
void R1csExport<FrP>::build(const R1CS<FrP>& cs) {
    const size_t nc = cs.constraint_count();
    for (int m = 0; m < 3; m++) {
        const auto& rows = cs.rows(m);
        ptr[m].assign(1, 0);
        col[m].clear();
        val[m].clear();
        ptr[m].resize(rows.size() + 1);
        size_t nnz = 0;
        for (size_t r = 0; r < rows.size(); r++) { nnz += rows[r].terms.size(); ptr[m][r + 1] = (uint32_t)nnz; }
        col[m].resize(nnz);
        val[m].resize(nnz * 4);
        uint32_t* colp = col[m].data();
        uint64_t* valp = val[m].data();
        const uint32_t* pp = ptr[m].data();
        parallel_chunks(rows.size(), 4096, [&, colp, valp, pp](size_t lo, size_t hi, size_t) {
            for (size_t r = lo; r < hi; r++) {
                size_t k = pp[r];
                // variable order: instance block then witness block; keys sort the same way (witness bit is the top bit)
                for (auto& t : rows[r].terms) {
                    colp[k] = cs.var_index(t.first);
                    to_canon_words<FrP>(valp + 4 * k, t.second);
                    k++;
                }
            }
        });
        view.row_ptr[m] = ptr[m].data();
        view.col[m] = col[m].data();
        view.val[m] = val[m].data();
    }
    const auto& inst = cs.instance_assignment();
    const auto& wit = cs.witness_assignment();
    assignment.resize((inst.size() + wit.size()) * 4);
    for (size_t i = 0; i < inst.size(); i++) to_canon_words<FrP>(&assignment[4 * i], inst[i]);
    uint64_t* asg = assignment.data() + 4 * inst.size();
    parallel_chunks(wit.size(), 8192, [&, asg](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; i++) to_canon_words<FrP>(asg + 4 * i, wit[i]);
    });
    view.n_constraints = (uint32_t)nc;
    view.n_instance = (uint32_t)inst.size();
    view.n_witness = (uint32_t)wit.size();
}

template <class FrP>
static Fp<FrP> sample_nonzero_mont(SplitMix64& rng, Fp<FrP>* canon_out) {
    for (;;) {
        Fp<FrP> c = sample_canonical<FrP>(rng);
        if (c.is_zero()) continue;
        if (canon_out) *canon_out = c;
        return zl::to_mont(c);
    }
}

// Groth16::compile -- ark_groth16::generate_random_parameters / generate_parameters (groth16.rs:427-443):
// evaluate the QAP at tau through the Lagrange coefficients, then fixed-base multiplications of the generators.  The
// fixed-base part runs on the device (zl_bases_generate); the trapdoor stays in the ProvingContext for exponent checks.
template <class E>
Result<std::pair<typename Groth16<E>::ProvingContext, typename Groth16<E>::VerifyingContext>> Groth16<E>::compile(zl_ctx* ctx, const Compiler& cs,
                                                                                                                  SplitMix64& rng, const R1csExport<FrP>* exported) {
    Result<std::pair<ProvingContext, VerifyingContext>> res{false, {}, Error{ZL_EINVAL}};
    if (!ctx) return res;
    const size_t nc = cs.constraint_count(), ni = cs.num_instance_variables(), nw = cs.secret_variable_count(), nv = ni + nw;
    unsigned log_n = 1;
    while (((size_t)1 << log_n) < nc + ni) log_n++;
    if (log_n > (unsigned)FrP::TWO_ADICITY) return res;
    const size_t N = (size_t)1 << log_n;
    Trapdoor td;
    const F alpha = sample_nonzero_mont<FrP>(rng, &td.alpha), beta = sample_nonzero_mont<FrP>(rng, &td.beta),
            gamma = sample_nonzero_mont<FrP>(rng, &td.gamma), delta = sample_nonzero_mont<FrP>(rng, &td.delta);
    F w;
    for (int i = 0; i < F::N; i++) w.l[i] = FrP::two_adic_root(i);
    for (unsigned i = log_n; i < (unsigned)FrP::TWO_ADICITY; i++) w = zl::sqr(w);
    // tau outside the domain
    F tau, zt;
    for (;;) {
        tau = sample_nonzero_mont<FrP>(rng, &td.tau);
        zt = tau;
        for (unsigned i = 0; i < log_n; i++) zt = zl::sqr(zt);
        zt = zl::sub(zt, F::one());  // Z(tau) = tau^N - 1
        if (!zt.is_zero()) break;
    }
    // Lagrange coefficients L_j = Z(tau)/N * w^j / (tau - w^j), batch inversion
    const bool dbg = getenv("ZL_DEBUG_TIMING") != nullptr;
    auto tmark = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[zl] compile: %-28s %.3f s\n", what, std::chrono::duration<double>(now - tmark).count());
        tmark = now;
    };
    std::vector<F> L(N), den(N), pref(N);
    {
        const F scale = zl::mul(zt, zl::inv(zl::from_u64<FrP>((uint64_t)N)));
        parallel_chunks(N, 4096, [&](size_t lo, size_t hi, size_t) {  // every chunk shares one inversion among its denominators
            F wj = pow_u64<FrP>(w, (uint64_t)lo);
            F acc = F::one();
            for (size_t j = lo; j < hi; j++) {
                den[j] = zl::sub(tau, wj);
                L[j] = wj;
                wj = zl::mul(wj, w);
                pref[j] = acc;
                acc = zl::mul(acc, den[j]);
            }
            F inv = zl::inv(acc);
            for (size_t j = hi; j-- > lo;) {
                const F dinv = zl::mul(inv, pref[j]);
                inv = zl::mul(inv, den[j]);
                L[j] = zl::mul(zl::mul(L[j], dinv), scale);
            }
        });
    }
    lap("lagrange");
    std::vector<F> u(nv, F::zero()), v(nv, F::zero()), ww(nv, F::zero());
    for (size_t j = 0; j < ni; j++) u[j] = L[nc + j];
    {
        // u = A^T L, v = B^T L, w = C^T L: scatter-adds over the variables.  Row chunks accumulate into private vectors (<= 8 per matrix),
        // which are then summed per variable range; the three matrices run side by side.
        auto transposed = [&](int m, std::vector<F>& dst) {
            const auto& rows = cs.rows(m);
            const size_t parts = std::min<size_t>(8, std::max<size_t>(1, rows.size() / 65536));
            std::vector<std::vector<F>> priv(parts > 1 ? parts - 1 : 0, std::vector<F>(nv, F::zero()));
            const size_t per = (rows.size() + parts - 1) / parts;
            std::vector<std::thread> th;
            for (size_t p = 0; p < parts; p++) {
                th.emplace_back([&, p]() {
                    std::vector<F>& acc = p == 0 ? dst : priv[p - 1];
                    const size_t lo = p * per, hi = std::min(rows.size(), lo + per);
                    for (size_t row = lo; row < hi; row++)
                        for (auto& t : rows[row].terms) {
                            const uint32_t i = cs.var_index(t.first);
                            acc[i] = zl::add(acc[i], zl::mul(t.second, L[row]));
                        }
                });
            }
            for (auto& x : th) x.join();
            if (parts > 1)
                parallel_chunks(nv, 8192, [&](size_t lo, size_t hi, size_t) {
                    for (auto& pv : priv)
                        for (size_t i = lo; i < hi; i++) dst[i] = zl::add(dst[i], pv[i]);
                });
        };
        std::thread tb([&]() { transposed(1, v); }), tc([&]() { transposed(2, ww); });
        transposed(0, u);
        tb.join();
        tc.join();
    }
    lap("transposed products");
    const F dinv = zl::inv(delta), ginv = zl::inv(gamma);
    // exponent vectors (canonical) for the device generator
    std::vector<uint64_t> ea(nv * 4), eb(nv * 4), eh((N - 1) * 4), el(std::max<size_t>(nw, 1) * 4), single(5 * 4);
    VerifyingContext vk;
    vk.gamma_abc_exponents.resize(ni);
    parallel_chunks(nv, 8192, [&](size_t lo, size_t hi, size_t) {
        for (size_t i = lo; i < hi; i++) {
            to_canon_words<FrP>(&ea[4 * i], u[i]);
            to_canon_words<FrP>(&eb[4 * i], v[i]);
            const F comb = zl::add(zl::add(zl::mul(beta, u[i]), zl::mul(alpha, v[i])), ww[i]);
            if (i < ni) vk.gamma_abc_exponents[i] = zl::mul(comb, ginv);
            else to_canon_words<FrP>(&el[4 * (i - ni)], zl::mul(comb, dinv));
        }
    });
    parallel_chunks(N - 1, 8192, [&](size_t lo, size_t hi, size_t) {
        F tp = zl::mul(zl::mul(zt, dinv), pow_u64<FrP>(tau, (uint64_t)lo));
        for (size_t j = lo; j < hi; j++) { to_canon_words<FrP>(&eh[4 * j], tp); tp = zl::mul(tp, tau); }
    });
    lap("exponent vectors");
    ProvingContext pc;
    pc.ctx = ctx;
    pc.trapdoor = td;
    pc.has_trapdoor = true;
    pc.n_instance = ni;
    pc.n_witness = nw;
    pc.domain_size = N;
    int rc = zl_bases_generate(ctx, E::curve, ZL_G1, ea.data(), nv, &pc.a_query);
    if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G1, eb.data(), nv, &pc.b_g1_query);
    if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G2, eb.data(), nv, &pc.b_g2_query);
    if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G1, eh.data(), N - 1, &pc.h_query);
    if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G1, el.data(), nw, &pc.l_query);
    lap("device generation (5 queries)");
    if (!rc) rc = build_window_tables(pc);
    lap("window tables");
    if (!rc) {
        R1csExport<FrP> own;
        if (!exported) own.build(cs);
        rc = zl_r1cs_upload(ctx, E::curve, exported ? &exported->view : &own.view, &pc.r1cs);
        pc.n_constraints = nc;
        pc.circuit_digest = cs.structure_digest();
        pc.witness_only_ok = cs.witness_only_compatible();
    }
    lap("r1cs export + upload");
    // alpha*G1, beta*G1, delta*G1, beta*G2, delta*G2
    const size_t q1 = 2 * E::G1::FQ64, q2 = 4 * E::G1::FQ64;
    pc.alpha_g1.resize(q1); pc.beta_g1.resize(q1); pc.delta_g1.resize(q1); pc.beta_g2.resize(q2); pc.delta_g2.resize(q2);
    if (!rc) {
        uint64_t k3[12];
        memcpy(k3, td.alpha.l, 32); memcpy(k3 + 4, td.beta.l, 32); memcpy(k3 + 8, td.delta.l, 32);
        uint64_t h1 = 0, h2 = 0;
        std::vector<uint64_t> tmp(3 * q2);
        rc = zl_bases_generate(ctx, E::curve, ZL_G1, k3, 3, &h1);
        if (!rc) rc = zl_bases_download(ctx, h1, 0, 3, tmp.data());
        if (!rc) {
            memcpy(pc.alpha_g1.data(), &tmp[0], q1 * 8);
            memcpy(pc.beta_g1.data(), &tmp[q1], q1 * 8);
            memcpy(pc.delta_g1.data(), &tmp[2 * q1], q1 * 8);
        }
        if (h1) (void)zl_bases_free(ctx, h1);
        if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G2, k3 + 4, 2, &h2);
        if (!rc) rc = zl_bases_download(ctx, h2, 0, 2, tmp.data());
        if (!rc) {
            memcpy(pc.beta_g2.data(), &tmp[0], q2 * 8);
            memcpy(pc.delta_g2.data(), &tmp[q2], q2 * 8);
        }
        if (h2) (void)zl_bases_free(ctx, h2);
    }
    // verifying key: alpha G1, beta G2, gamma G2, delta G2, gamma_abc_i G1 (device fixed-base, then copied back)
    if (!rc) {
        vk.alpha_g1 = pc.alpha_g1;
        vk.beta_g2 = pc.beta_g2;
        vk.delta_g2 = pc.delta_g2;
        vk.gamma_g2.resize(q2);
        vk.gamma_abc_g1.resize(ni * q1);
        uint64_t kg[4];
        memcpy(kg, td.gamma.l, 32);
        uint64_t h = 0;
        rc = zl_bases_generate(ctx, E::curve, ZL_G2, kg, 1, &h);
        if (!rc) rc = zl_bases_download(ctx, h, 0, 1, vk.gamma_g2.data());
        if (h) (void)zl_bases_free(ctx, h);
        std::vector<uint64_t> eg(ni * 4);
        for (size_t i = 0; i < ni; i++) to_canon_words<FrP>(&eg[4 * i], vk.gamma_abc_exponents[i]);
        h = 0;
        if (!rc) rc = zl_bases_generate(ctx, E::curve, ZL_G1, eg.data(), ni, &h);
        if (!rc) rc = zl_bases_download(ctx, h, 0, ni, vk.gamma_abc_g1.data());
        if (h) (void)zl_bases_free(ctx, h);
    }
    lap("single points + vk");
    if (rc) {
        release(pc);
        res.error = Error{rc};
        return res;
    }
    res.ok = true;
    res.value = {pc, vk};
    return res;
}

template <class E>
Result<bool> Groth16<E>::verify(const VerifyingContext& vk, const Input& input, const Proof& proof) {
    using G1 = typename E::G1;
    using F1 = typename G1::F;
    using Eng = openzl::pairing::Engine<typename G1::FqP, typename E::PairingP>;
    Result<bool> res{false, false, Error{ZL_EINVAL}};
    const size_t q1 = 2 * G1::FQ64;
    const size_t ni = vk.gamma_abc_g1.size() / q1;
    if (ni == 0 || input.size() + 1 != ni) return res;  // ark: wrong number of public inputs -> Err(MalformedVerifyingKey)
    constexpr int W1 = FieldIO<F1>::WORDS;
    auto load = [&](const uint64_t* xy) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(xy);
        uint32_t acc = 0;
        for (int k = 0; k < 2 * W1; k++) acc |= w[k];
        if (!acc) return XYZZ<F1>::inf();
        return XYZZ<F1>::from_affine(Affine<F1>{FieldIO<F1>::load_canon(w), FieldIO<F1>::load_canon(w + W1)});
    };
    // prepare_inputs: g_ic = gamma_abc[0] + sum_i x_i gamma_abc[i + 1]
    XYZZ<F1> acc = load(vk.gamma_abc_g1.data());
    for (size_t i = 0; i < input.size(); i++) {
        uint32_t k[8];
        memcpy(k, input[i].l, 32);
        zl::add_full(acc, zl::mul_scalar(load(vk.gamma_abc_g1.data() + (i + 1) * q1), k));
    }
    auto neg_canon = [&](const XYZZ<F1>& p, std::vector<uint64_t>& out) {  // canonical affine of -p
        out.assign(q1, 0);
        if (p.is_inf()) return;
        XYZZ<F1> n = p;
        zl::neg_inplace(n);
        const Affine<F1> a = zl::to_affine(n);
        uint32_t* w = reinterpret_cast<uint32_t*>(out.data());
        FieldIO<F1>::store_canon(w, a.x);
        FieldIO<F1>::store_canon(w + W1, a.y);
    };
    std::vector<uint64_t> n_alpha, n_acc, n_c;
    neg_canon(load(vk.alpha_g1.data()), n_alpha);
    neg_canon(acc, n_acc);
    neg_canon(load(proof.c), n_c);
    // e(A, B) e(-alpha, beta) e(-g_ic, gamma) e(-C, delta) == 1, one final exponentiation
    const auto gt = Eng::multi_pairing({proof.a, n_alpha.data(), n_acc.data(), n_c.data()},
                                       {proof.b, vk.beta_g2.data(), vk.gamma_g2.data(), vk.delta_g2.data()});
    res.ok = true;
    res.value = !proof.a_inf && !proof.b_inf && Eng::eq(gt, Eng::one());
    return res;
}

// Window tables for the queries of a large proving key (static per circuit, like the key itself): the five MSMs of every proof then
// run on one merged bucket set each.  Measured at N = 2^20 (958 465 constraints): prove 33.5 -> 29.2 ms in round 1 with c = 20 for the G1
// queries (13 windows) and c = 16 for the G2 query; re-measured in round 3 (19.4 ms): G1 c = 19 / 21 -> 20.9 / 22.6 ms, G2 c = 17 / 18 / 20 -> 19.4 / 19.2 / 20.5;
// round 5 (18.1 ms, profiles/r05_g16_table_widths.log, two interleaved passes): G1 19 / 21 -> 19.4 / 20.6, G2 17 / 18 / 19 -> 18.4 / 17.9-18.35 / 19.0: the defaults stand.
// Memory: 13 x 128 B per G1 point, 16 x 256 B per G2 point (about 11 GB for this key).  Small keys stay on the plain path.
template <class E>
int Groth16<E>::build_window_tables(const ProvingContext& pc) {
    const size_t big = (size_t)1 << 19, nv = pc.n_instance + pc.n_witness;
    const char* e1 = getenv("ZL_TUNE_G1_TABLE_C");
    const char* e2 = getenv("ZL_TUNE_G2_TABLE_C");
    const int c1 = e1 ? atoi(e1) : 20, c2 = e2 ? atoi(e2) : 16;
    const struct { uint64_t h; size_t n; int c; } q[5] = {
        {pc.a_query, nv, c1}, {pc.b_g1_query, nv, c1}, {pc.h_query, pc.domain_size - 1, c1}, {pc.l_query, pc.n_witness, c1}, {pc.b_g2_query, nv, c2}};
    int rc = ZL_OK;
    for (const auto& e : q)
        if (!rc && e.n >= big) rc = zl_bases_precompute(pc.ctx, e.h, e.c);
    return rc;
}

template <class E> struct CodecOf;
template <> struct CodecOf<Bls12_381> { using type = serialize::BlsCodec; };
template <> struct CodecOf<Bn254> { using type = serialize::BnCodec; };

static void put_u64le(std::vector<uint8_t>& out, uint64_t v) {
    for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i)));
}
static bool all_zero_words(const uint64_t* w, size_t n) {
    uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) acc |= w[i];
    return acc == 0;
}

template <class E>
std::vector<uint8_t> Groth16<E>::encode_verifying_key(const VerifyingContext& vk) {
    using C = typename CodecOf<E>::type;
    const size_t q1 = 2 * E::G1::FQ64, ni = vk.gamma_abc_g1.size() / q1;
    std::vector<uint8_t> out(C::G1_BYTES + 3 * C::G2_BYTES);
    uint8_t* p = out.data();
    C::g1_to_bytes(vk.alpha_g1.data(), all_zero_words(vk.alpha_g1.data(), q1), p); p += C::G1_BYTES;
    for (const auto* g : {&vk.beta_g2, &vk.gamma_g2, &vk.delta_g2}) { C::g2_to_bytes(g->data(), all_zero_words(g->data(), 2 * q1), p); p += C::G2_BYTES; }
    put_u64le(out, ni);
    const size_t at = out.size();
    out.resize(at + ni * C::G1_BYTES);
    for (size_t i = 0; i < ni; i++) C::g1_to_bytes(&vk.gamma_abc_g1[i * q1], all_zero_words(&vk.gamma_abc_g1[i * q1], q1), out.data() + at + i * C::G1_BYTES);
    return out;
}

template <class E>
Result<std::vector<uint8_t>> Groth16<E>::encode(const ProvingContext& pc, const VerifyingContext& vk) {
    using C = typename CodecOf<E>::type;
    Result<std::vector<uint8_t>> res{false, {}, Error{ZL_EINVAL}};
    if (!pc.ctx) return res;
    const size_t q1 = 2 * E::G1::FQ64, q2 = 2 * q1, ni = pc.n_instance, nv = pc.n_instance + pc.n_witness;
    if (vk.gamma_abc_g1.size() != ni * q1 || pc.domain_size < 2) return res;
    std::vector<uint8_t>& out = res.value;
    auto put_g1 = [&](const uint64_t* xy) { const size_t at = out.size(); out.resize(at + C::G1_UNC_BYTES); C::g1_to_uncompressed(xy, all_zero_words(xy, q1), out.data() + at); };
    auto put_g2 = [&](const uint64_t* xy) { const size_t at = out.size(); out.resize(at + C::G2_UNC_BYTES); C::g2_to_uncompressed(xy, all_zero_words(xy, q2), out.data() + at); };
    // vk
    put_g1(vk.alpha_g1.data()); put_g2(vk.beta_g2.data()); put_g2(vk.gamma_g2.data()); put_g2(vk.delta_g2.data());
    put_u64le(out, ni);
    for (size_t i = 0; i < ni; i++) put_g1(&vk.gamma_abc_g1[i * q1]);
    put_g1(pc.beta_g1.data()); put_g1(pc.delta_g1.data());
    // the five queries: downloaded as canonical affine words, re-encoded in place (only infinity records differ from the raw words)
    const struct { uint64_t h; size_t n; bool g2; } q[5] = {
        {pc.a_query, nv, false}, {pc.b_g1_query, nv, false}, {pc.b_g2_query, nv, true}, {pc.h_query, pc.domain_size - 1, false}, {pc.l_query, pc.n_witness, false}};
    for (const auto& e : q) {
        put_u64le(out, e.n);
        const size_t rec = e.g2 ? C::G2_UNC_BYTES : C::G1_UNC_BYTES, words = rec / 8, at = out.size();
        out.resize(at + e.n * rec);
        if (!e.n) continue;
        std::vector<uint64_t> tmp(e.n * words);
        const int rc = zl_bases_download(pc.ctx, e.h, 0, e.n, tmp.data());
        if (rc) { res.error = Error{rc}; res.value.clear(); return res; }
        uint8_t* base = out.data() + at;
        parallel_chunks(e.n, 16384, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; i++) {
                const uint64_t* xy = &tmp[i * words];
                if (e.g2) C::g2_to_uncompressed(xy, all_zero_words(xy, words), base + i * rec);
                else C::g1_to_uncompressed(xy, all_zero_words(xy, words), base + i * rec);
            }
        });
    }
    res.ok = true;
    return res;
}

template <class E>
Result<std::pair<typename Groth16<E>::ProvingContext, typename Groth16<E>::VerifyingContext>> Groth16<E>::decode(zl_ctx* ctx, const uint8_t* in, size_t len,
                                                                                                                   unsigned flags) {
    using C = typename CodecOf<E>::type;
    Result<std::pair<ProvingContext, VerifyingContext>> res{false, {}, Error{ZL_EINVAL}};
    if (!in) return res;  // ctx == nullptr: PARSE ONLY (zl_groth16_keys_parse) -- every byte is read and validated exactly as below, nothing goes to a device
    const size_t q1 = 2 * E::G1::FQ64, q2 = 2 * q1;
    const bool check = (flags & ZL_CHECK) != 0;
    size_t pos = 0;
    int rc = ZL_OK;
    auto take = [&](size_t n) -> const uint8_t* {  // nullptr: the input ends early
        if (rc || n > len - pos) { if (!rc) rc = ZL_EINVAL; return nullptr; }
        const uint8_t* p = in + pos;
        pos += n;
        return p;
    };
    auto get_g1 = [&](std::vector<uint64_t>& xy) {
        xy.assign(q1, 0);
        uint8_t inf = 0;
        if (const uint8_t* p = take(C::G1_UNC_BYTES)) rc = C::g1_from_uncompressed(p, check, xy.data(), &inf);
    };
    auto get_g2 = [&](std::vector<uint64_t>& xy) {
        xy.assign(q2, 0);
        uint8_t inf = 0;
        if (const uint8_t* p = take(C::G2_UNC_BYTES)) rc = C::g2_from_uncompressed(p, check, xy.data(), &inf);
    };
    auto get_len = [&](size_t rec) -> size_t {  // Vec<T> length prefix, bounded by what the remaining input can hold
        const uint8_t* p = take(8);
        if (!p) return 0;
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)p[i] << (8 * i);
        if (v > (len - pos) / rec) { rc = ZL_EINVAL; return 0; }
        return (size_t)v;
    };
    ProvingContext& pc = res.value.first;
    VerifyingContext& vk = res.value.second;
    pc.ctx = ctx;
    get_g1(vk.alpha_g1); get_g2(vk.beta_g2); get_g2(vk.gamma_g2); get_g2(vk.delta_g2);
    const size_t ni = get_len(C::G1_UNC_BYTES);
    vk.gamma_abc_g1.assign(ni * q1, 0);
    for (size_t i = 0; i < ni && !rc; i++) {
        std::vector<uint64_t> t;
        get_g1(t);
        memcpy(&vk.gamma_abc_g1[i * q1], t.data(), q1 * 8);
    }
    get_g1(pc.beta_g1); get_g1(pc.delta_g1);
    pc.alpha_g1 = vk.alpha_g1;
    pc.beta_g2 = vk.beta_g2;
    pc.delta_g2 = vk.delta_g2;
    // the five queries: validated and converted to the all-zero infinity encoding on the host, then uploaded
    size_t count[5] = {0, 0, 0, 0, 0};
    uint64_t* handle[5] = {&pc.a_query, &pc.b_g1_query, &pc.b_g2_query, &pc.h_query, &pc.l_query};
    for (int k = 0; k < 5 && !rc; k++) {
        const bool g2 = k == 2;
        const size_t rec = g2 ? C::G2_UNC_BYTES : C::G1_UNC_BYTES, words = rec / 8;
        const size_t n = count[k] = get_len(rec);
        const uint8_t* p = take(n * rec);
        if (rc) break;
        if (n == 0) continue;  // an empty query (a circuit without witnesses) has no handle; prove() never reads it
        std::vector<uint64_t> xy(n * words);
        std::atomic<int> bad{ZL_OK};
        parallel_chunks(n, 16384, [&](size_t lo, size_t hi, size_t) {
            for (size_t i = lo; i < hi; i++) {
                uint8_t inf = 0;
                // subgroup membership on the host would cost a scalar multiplication per point: with ZL_CHECK the device verifies the
                // curve equation of every uploaded point instead (zl_bases_upload)
                const int r = g2 ? C::g2_from_uncompressed(p + i * rec, false, &xy[i * words], &inf) : C::g1_from_uncompressed(p + i * rec, false, &xy[i * words], &inf);
                if (r) bad.store(r);
            }
        });
        if ((rc = bad.load())) break;
        if (ctx) rc = zl_bases_upload(ctx, E::curve, g2 ? ZL_G2 : ZL_G1, xy.data(), n, 0, -1, ZL_CANON | (check ? ZL_CHECK : 0u), handle[k]);
    }
    if (!rc && pos != len) rc = ZL_EINVAL;  // trailing bytes
    // shape: a, b1, b2 cover every variable; h has N - 1 entries for a power-of-two domain N >= 2; l covers the witnesses
    if (!rc) {
        const size_t nv = count[0], N = count[3] + 1;
        if (ni < 1 || nv < ni || count[1] != nv || count[2] != nv || count[4] != nv - ni || N < 2 || (N & (N - 1))) rc = ZL_EINVAL;
        pc.n_instance = ni;
        pc.n_witness = nv - ni;
        pc.domain_size = N;
    }
    if (!rc && ctx) rc = build_window_tables(pc);
    if (rc) {
        release(pc);
        res.value = {};
        res.error = Error{rc};
        return res;
    }
    res.ok = true;
    return res;
}

template <class E>
void Groth16<E>::release(ProvingContext& pc) {
    if (!pc.ctx) return;
    // a handle whose free is REFUSED (a fork the caller made still lives: ZL_EINVAL) stays in the context, so that a later release -- or zl_ctx_destroy --
    // still finds it (zeroing it leaked the device-resident key)
    for (uint64_t* h : {&pc.a_query, &pc.b_g1_query, &pc.b_g2_query, &pc.h_query, &pc.l_query})
        if (*h && zl_bases_free(pc.ctx, *h) != ZL_EINVAL) *h = 0;
    if (pc.r1cs && zl_r1cs_free(pc.ctx, pc.r1cs) != ZL_EINVAL) pc.r1cs = 0;
}

// Groth16::prove (groth16.rs:445-457): r, s <- rng; create_proof_with_assignment on the device
template <class E>
Result<typename Groth16<E>::Proof> Groth16<E>::prove(const ProvingContext& pc, const Compiler& cs, SplitMix64& rng, F* r_out, F* s_out, zl_ctx* lane) {
    Result<Proof> res{false, {}, Error{ZL_EINVAL}};
    if (!lane) lane = pc.ctx;
    if (lane != pc.ctx && (lane->parent != pc.ctx || !pc.r1cs)) return res;  // not a lane of this context / binding happens on the context's own ctx
    if (cs.mode() == Compiler::Mode::Setup) return res;  // the reference panics when a setup-mode compiler reaches prove
    if (cs.num_instance_variables() != pc.n_instance || cs.secret_variable_count() != pc.n_witness) return res;
    if (cs.witness_only() && !pc.r1cs) return res;  // a witness-only compiler has no rows to bind an unbound context to
    if (cs.witness_only() && !pc.witness_only_ok) return res;  // the bound circuit allocates witnesses differently without its rows (R1CS::mul)
    const F r = sample_canonical<FrP>(rng), s = sample_canonical<FrP>(rng);
    if (r_out) *r_out = r;
    if (s_out) *s_out = s;
    if (pc.r1cs && !cs.witness_only() && (cs.constraint_count() != pc.n_constraints || cs.structure_digest() != pc.circuit_digest)) return res;  // not the bound circuit
    if (!pc.r1cs) {
        // a context decoded from bytes (Groth16::decode) does not know its circuit: the first proof uploads the matrices, after checking
        // that the compiler's evaluation domain is the one the key's h_query was generated for
        const size_t nc = cs.constraint_count();
        size_t N = 2;
        while (N < nc + pc.n_instance) N <<= 1;
        if (N != pc.domain_size) return res;
        R1csExport<FrP> ex;
        ex.build(cs);
        const int rc_up = zl_r1cs_upload(pc.ctx, E::curve, &ex.view, &pc.r1cs);
        if (rc_up) { res.error = Error{rc_up}; return res; }
        pc.n_constraints = nc;
        pc.circuit_digest = cs.structure_digest();
        pc.witness_only_ok = cs.witness_only_compatible();
    }
    // only the assignment travels per proof; the matrices and the proving key are device-resident.  It goes to the device straight from the
    // compiler's two vectors (Montgomery limbs as held): no host-side concatenation
    const auto& inst = cs.instance_assignment();
    const auto& wit = cs.witness_assignment();
    zl_g16_pk pk{};
    pk.curve = E::curve;
    pk.a_query = pc.a_query; pk.b_g1_query = pc.b_g1_query; pk.h_query = pc.h_query; pk.l_query = pc.l_query; pk.b_g2_query = pc.b_g2_query;
    pk.alpha_g1 = pc.alpha_g1.data(); pk.beta_g1 = pc.beta_g1.data(); pk.delta_g1 = pc.delta_g1.data();
    pk.beta_g2 = pc.beta_g2.data(); pk.delta_g2 = pc.delta_g2.data();
    uint64_t rw[4], sw[4];
    memcpy(rw, r.l, 32);
    memcpy(sw, s.l, 32);
    const int rc = zl_groth16_prove_split(lane, &pk, pc.r1cs, reinterpret_cast<const uint64_t*>(inst.data()),
                                          wit.empty() ? nullptr : reinterpret_cast<const uint64_t*>(wit.data()), rw, sw, &res.value);
    if (rc) { res.error = Error{rc}; return res; }  // .map_err(|_| Error) groth16.rs:456
    res.ok = true;
    return res;
}

template struct Groth16<Bls12_381>;
template struct Groth16<Bn254>;
template struct R1csExport<BLS12_381_Fr>;
template struct R1csExport<BN254_Fr>;

// config 5 circuit: h_1 = H(x0, x1), h_{j+1} = H(h_j, x1), public input h_k  
// the same circuit code run by a witness-only compiler (R1CS::for_witness): every link through the gadget, values only
template <class FrP>
static R1CS<FrP> poseidon_chain_witness(uint32_t k, const Fp<FrP>& x0_canon, const Fp<FrP>& x1_canon) {
    static const poseidon::Constants<FrP> consts;
    R1CS<FrP> cs = R1CS<FrP>::for_witness();
    FpVar<FrP> cur = cs.new_witness(zl::to_mont(x0_canon));
    const FpVar<FrP> b = cs.new_witness(zl::to_mont(x1_canon));
    for (uint32_t j = 0; j < k; j++) cur = poseidon::hash(consts, cur, b, cs);
    // the public input is the chain's output (the instance block is its own vector: allocating it last changes no index)
    const FpVar<FrP> out_pub = cs.new_input(cur.value);
    cs.enforce_equal(cur, out_pub);
    return cs;
}
template <class FrP>
static R1CS<FrP> poseidon_chain(uint32_t k, const Fp<FrP>& x0_canon, const Fp<FrP>& x1_canon) {
    using F = Fp<FrP>;
    static const poseidon::Constants<FrP> consts;
    R1CS<FrP> cs = R1CS<FrP>::for_proofs();
    // expected output natively (public input is allocated first so that the instance block precedes the witnesses)
    F h = zl::to_mont(x0_canon);
    const F x1 = zl::to_mont(x1_canon);
    for (uint32_t j = 0; j < k; j++) {
        F st[3] = {zl::from_u64<FrP>(3), h, x1};
        poseidon::permute_native(consts, st);
        h = st[0];
    }
    FpVar<FrP> out_pub = cs.new_input(h);
    FpVar<FrP> a = cs.new_witness(zl::to_mont(x0_canon));
    FpVar<FrP> b = cs.new_witness(x1);
    FpVar<FrP> cur = a;
    // Links 1 and 2 are synthesised symbolically (link 1 takes a plain variable, every later link the previous link's output combination);
    // links 3.. are copies of link 2's constraints with the witness indices shifted by one link each, their witness values recomputed
    // natively.  Same circuit as k symbolic syntheses (the linear-combination arithmetic of a link is ~10^4 field multiplications and
    // was 2.5 s of host time at k = 4096).
    const uint32_t sym = k < 2 ? k : 2;
    const auto t0 = std::chrono::steady_clock::now();
    size_t rows0 = 0, wit0 = 0;
    F hv = zl::to_mont(x0_canon);
    for (uint32_t j = 0; j < sym; j++) {
        rows0 = cs.constraint_count();
        wit0 = cs.secret_variable_count();
        cur = poseidon::hash(consts, cur, b, cs);
        hv = cur.value;
    }
    const auto tA = std::chrono::steady_clock::now();
    if (getenv("ZL_DEBUG_TIMING")) fprintf(stderr, "[zl] chain: native chain + 2 symbolic links %.3f s\n", std::chrono::duration<double>(tA - t0).count());
    if (k > sym) {
        const size_t rows1 = cs.constraint_count(), per_link = cs.secret_variable_count() - wit0, copies = k - sym;
        std::vector<F> vals;
        vals.reserve(per_link * copies);
        for (size_t j = 0; j < copies; j++) {
            F st[3] = {zl::from_u64<FrP>(3), hv, x1};
            poseidon::permute_native_record(consts, st, vals);
            hv = st[0];
        }
        const auto tB = std::chrono::steady_clock::now();
        // every witness of links 1 and 2 (index >= 2: after x0, x1) moves with the copy; the instance block, x0 and x1 stay
        cs.replicate_rows(rows0, rows1, copies, 2, (uint32_t)per_link, vals);
        if (getenv("ZL_DEBUG_TIMING"))
            fprintf(stderr, "[zl] chain: native values %.3f s, replicate %.3f s\n", std::chrono::duration<double>(tB - tA).count(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tB).count());
        cur.lc = R1CS<FrP>::shifted(cur.lc, 2, (uint32_t)(copies * per_link));
        cur.value = hv;
    }
    cs.enforce_equal(cur, out_pub);
    return cs;
}

}  // namespace openzl

// ------------------------------------------------------------------------------------------------ C-ABI hooks
using namespace openzl;
struct zl_circuit {
    zl_curve_t curve;
    R1CS<BLS12_381_Fr>* bls = nullptr;
    R1CS<BN254_Fr>* bn = nullptr;
    R1csExport<BLS12_381_Fr> ex_bls;
    R1csExport<BN254_Fr> ex_bn;
};
// include/zl_backend_test.h: turn the circuit into a DIFFERENT circuit of the same shape (first coefficient of A's row 0 doubled); its
// cached CSR export is dropped.  A proving context bound to the original circuit must refuse it (R1CS::structure_digest).
extern "C" int zl_test_circuit_tweak(zl_circuit* c) {
    if (!c) return ZL_EINVAL;
    if (c->bls) { c->bls->tweak_for_tests(); c->ex_bls = R1csExport<BLS12_381_Fr>(); c->ex_bls.build(*c->bls); }
    if (c->bn) { c->bn->tweak_for_tests(); c->ex_bn = R1csExport<BN254_Fr>(); c->ex_bn.build(*c->bn); }
    return ZL_OK;
}
struct zl_g16_keys {
    zl_curve_t curve;
    Groth16<Bls12_381>::ProvingContext pc_bls;
    Groth16<Bn254>::ProvingContext pc_bn;
    std::vector<uint64_t> gamma_abc;  // exponents, canonical
    Groth16<Bls12_381>::VerifyingContext vk_bls;
    Groth16<Bn254>::VerifyingContext vk_bn;
};

template <class E> static const R1CS<typename E::FrP>* cs_of(const zl_circuit* c);
template <> const R1CS<BLS12_381_Fr>* cs_of<Bls12_381>(const zl_circuit* c) { return c->bls; }
template <> const R1CS<BN254_Fr>* cs_of<Bn254>(const zl_circuit* c) { return c->bn; }
// A stream of proofs over one key on two prover lanes (include/zl_backend.h): two host threads draw indices from one counter; thread 0 is the caller's
template <class E>
static int prove_circuits_t(zl_ctx* ctx, const typename Groth16<E>::ProvingContext& pc, const zl_circuit* const* circuits, const uint64_t* seeds, size_t count,
                            zl_g16_proof* proofs) {
    if (pc.ctx != ctx || ctx->parent || !pc.r1cs) return ZL_EINVAL;
    if (count > 1 && !ctx->stream_lane_ctx) {
        const int rc = zl_ctx_fork(ctx, &ctx->stream_lane_ctx);
        if (rc) return rc;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> first_err{ZL_OK};
    auto work = [&](zl_ctx* lane) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= count || first_err.load() != ZL_OK) return;
            const zl_circuit* c = circuits[i];
            int rc = ZL_EINVAL;
            if (c && c->curve == E::curve) {
                SplitMix64 rng(seeds[i]);
                const auto* cs = cs_of<E>(c);
                auto res = Groth16<E>::prove(pc, *cs, rng, nullptr, nullptr, lane);
                rc = res.ok ? ZL_OK : res.error.code;
                if (res.ok) proofs[i] = res.value;
            }
            if (rc != ZL_OK) { int exp = ZL_OK; first_err.compare_exchange_strong(exp, rc); return; }
        }
    };
    if (count > 1) {
        std::thread second(work, ctx->stream_lane_ctx);
        work(ctx);
        second.join();
    } else {
        work(ctx);
    }
    return first_err.load();
}

extern "C" {

int zl_circuit_poseidon_chain(zl_curve_t curve, uint32_t k, const uint64_t* x0, const uint64_t* x1, zl_circuit** out) {
    if (!out || !x0 || !x1 || k == 0 || (curve != ZL_BLS12_381 && curve != ZL_BN254)) return ZL_EINVAL;
    zl_circuit* c = new (std::nothrow) zl_circuit();
    if (!c) return ZL_ENOMEM;
    c->curve = curve;
    if (curve == ZL_BLS12_381) {
        Fp<BLS12_381_Fr> a, b;
        memcpy(a.l, x0, 32); memcpy(b.l, x1, 32);
        const auto t0 = std::chrono::steady_clock::now();
        c->bls = new R1CS<BLS12_381_Fr>(poseidon_chain<BLS12_381_Fr>(k, a, b));
        const auto t1 = std::chrono::steady_clock::now();
        c->ex_bls.build(*c->bls);
        if (getenv("ZL_DEBUG_TIMING"))
            fprintf(stderr, "[zl] poseidon_chain %.3f s, export %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    } else {
        Fp<BN254_Fr> a, b;
        memcpy(a.l, x0, 32); memcpy(b.l, x1, 32);
        c->bn = new R1CS<BN254_Fr>(poseidon_chain<BN254_Fr>(k, a, b));
        c->ex_bn.build(*c->bn);
    }
    *out = c;
    return ZL_OK;
}
// the same circuit synthesised by a witness-only compiler (values, no rows): for proofs against a context that already holds the circuit's matrices
int zl_circuit_poseidon_chain_witness(zl_curve_t curve, uint32_t k, const uint64_t* x0, const uint64_t* x1, zl_circuit** out) {
    if (!out || !x0 || !x1 || k == 0 || (curve != ZL_BLS12_381 && curve != ZL_BN254)) return ZL_EINVAL;
    zl_circuit* c = new (std::nothrow) zl_circuit();
    if (!c) return ZL_ENOMEM;
    c->curve = curve;
    if (curve == ZL_BLS12_381) {
        Fp<BLS12_381_Fr> a, b;
        memcpy(a.l, x0, 32); memcpy(b.l, x1, 32);
        c->bls = new R1CS<BLS12_381_Fr>(poseidon_chain_witness<BLS12_381_Fr>(k, a, b));
        c->ex_bls.build(*c->bls);  // (no rows: only the assignment is exported)
    } else {
        Fp<BN254_Fr> a, b;
        memcpy(a.l, x0, 32); memcpy(b.l, x1, 32);
        c->bn = new R1CS<BN254_Fr>(poseidon_chain_witness<BN254_Fr>(k, a, b));
        c->ex_bn.build(*c->bn);
    }
    *out = c;
    return ZL_OK;
}
void zl_circuit_free(zl_circuit* c) {
    if (!c) return;
    delete c->bls;
    delete c->bn;
    delete c;
}
int zl_circuit_export(const zl_circuit* c, zl_r1cs* view, const uint64_t** assignment) {
    if (!c || !view) return ZL_EINVAL;
    if (c->curve == ZL_BLS12_381) { *view = c->ex_bls.view; if (assignment) *assignment = c->ex_bls.assignment.data(); }
    else { *view = c->ex_bn.view; if (assignment) *assignment = c->ex_bn.assignment.data(); }
    return ZL_OK;
}
int zl_circuit_is_satisfied(const zl_circuit* c) {
    if (!c) return ZL_EINVAL;
    return c->curve == ZL_BLS12_381 ? (c->bls->is_satisfied() ? 1 : 0) : (c->bn->is_satisfied() ? 1 : 0);
}
int zl_poseidon_permute(zl_curve_t curve, uint64_t* state) {
    if (!state) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) {
        static const poseidon::Constants<BLS12_381_Fr> cst;
        Fp<BLS12_381_Fr> st[3];
        for (int i = 0; i < 3; i++) { memcpy(st[i].l, state + 4 * i, 32); st[i] = zl::to_mont(st[i]); }
        poseidon::permute_native(cst, st);
        for (int i = 0; i < 3; i++) { st[i] = zl::from_mont(st[i]); memcpy(state + 4 * i, st[i].l, 32); }
        return ZL_OK;
    }
    if (curve == ZL_BN254) {
        static const poseidon::Constants<BN254_Fr> cst;
        Fp<BN254_Fr> st[3];
        for (int i = 0; i < 3; i++) { memcpy(st[i].l, state + 4 * i, 32); st[i] = zl::to_mont(st[i]); }
        poseidon::permute_native(cst, st);
        for (int i = 0; i < 3; i++) { st[i] = zl::from_mont(st[i]); memcpy(state + 4 * i, st[i].l, 32); }
        return ZL_OK;
    }
    return ZL_EINVAL;
}
int zl_groth16_compile(zl_ctx* ctx, const zl_circuit* c, uint64_t seed, zl_g16_keys** out) {
    if (!ctx || !c || !out) return ZL_EINVAL;
    zl_g16_keys* k = new (std::nothrow) zl_g16_keys();
    if (!k) return ZL_ENOMEM;
    k->curve = c->curve;
    SplitMix64 rng(seed);
    int rc = ZL_OK;
    if (c->curve == ZL_BLS12_381) {
        auto r = Groth16<Bls12_381>::compile(ctx, *c->bls, rng, &c->ex_bls);
        if (!r.ok) rc = r.error.code; else {
            k->pc_bls = r.value.first;
            k->vk_bls = r.value.second;
            for (auto& g : r.value.second.gamma_abc_exponents) { uint64_t w[4]; to_canon_words<BLS12_381_Fr>(w, g); k->gamma_abc.insert(k->gamma_abc.end(), w, w + 4); }
        }
    } else {
        auto r = Groth16<Bn254>::compile(ctx, *c->bn, rng, &c->ex_bn);
        if (!r.ok) rc = r.error.code; else {
            k->pc_bn = r.value.first;
            k->vk_bn = r.value.second;
            for (auto& g : r.value.second.gamma_abc_exponents) { uint64_t w[4]; to_canon_words<BN254_Fr>(w, g); k->gamma_abc.insert(k->gamma_abc.end(), w, w + 4); }
        }
    }
    if (rc) { delete k; return rc; }
    *out = k;
    return ZL_OK;
}
void zl_groth16_keys_free(zl_g16_keys* k) {
    if (!k) return;
    Groth16<Bls12_381>::release(k->pc_bls);
    Groth16<Bn254>::release(k->pc_bn);
    delete k;
}
int zl_groth16_keys_pk(const zl_g16_keys* k, zl_g16_pk* pk) {
    if (!k || !pk) return ZL_EINVAL;
    pk->curve = k->curve;
#define FILL(pc)                                                                                                     \
    pk->a_query = pc.a_query; pk->b_g1_query = pc.b_g1_query; pk->h_query = pc.h_query; pk->l_query = pc.l_query;    \
    pk->b_g2_query = pc.b_g2_query; pk->alpha_g1 = pc.alpha_g1.data(); pk->beta_g1 = pc.beta_g1.data();              \
    pk->delta_g1 = pc.delta_g1.data(); pk->beta_g2 = pc.beta_g2.data(); pk->delta_g2 = pc.delta_g2.data();
    if (k->curve == ZL_BLS12_381) { FILL(k->pc_bls) } else { FILL(k->pc_bn) }
#undef FILL
    return ZL_OK;
}
int zl_groth16_keys_trapdoor(const zl_g16_keys* k, uint64_t* out20) {
    if (!k || !out20) return ZL_EINVAL;
    if (!(k->curve == ZL_BLS12_381 ? k->pc_bls.has_trapdoor : k->pc_bn.has_trapdoor)) return ZL_EINVAL;  // keys decoded from bytes carry none
    if (k->curve == ZL_BLS12_381) {
        const auto& t = k->pc_bls.trapdoor;
        const Fp<BLS12_381_Fr>* v[5] = {&t.alpha, &t.beta, &t.gamma, &t.delta, &t.tau};
        for (int i = 0; i < 5; i++) memcpy(out20 + 4 * i, v[i]->l, 32);
    } else {
        const auto& t = k->pc_bn.trapdoor;
        const Fp<BN254_Fr>* v[5] = {&t.alpha, &t.beta, &t.gamma, &t.delta, &t.tau};
        for (int i = 0; i < 5; i++) memcpy(out20 + 4 * i, v[i]->l, 32);
    }
    return ZL_OK;
}
int zl_groth16_prove_circuit(zl_ctx* ctx, const zl_g16_keys* k, const zl_circuit* c, uint64_t seed, zl_g16_proof* proof, uint64_t* r_out, uint64_t* s_out) {
    if (!ctx || !k || !c || !proof || k->curve != c->curve) return ZL_EINVAL;
    SplitMix64 rng(seed);
    if (c->curve == ZL_BLS12_381) {
        Fp<BLS12_381_Fr> r, s;
        auto res = Groth16<Bls12_381>::prove(k->pc_bls, *c->bls, rng, &r, &s, ctx);
        if (!res.ok) return res.error.code;
        *proof = res.value;
        if (r_out) memcpy(r_out, r.l, 32);
        if (s_out) memcpy(s_out, s.l, 32);
    } else {
        Fp<BN254_Fr> r, s;
        auto res = Groth16<Bn254>::prove(k->pc_bn, *c->bn, rng, &r, &s, ctx);
        if (!res.ok) return res.error.code;
        *proof = res.value;
        if (r_out) memcpy(r_out, r.l, 32);
        if (s_out) memcpy(s_out, s.l, 32);
    }
    return ZL_OK;
}

int zl_groth16_prove_circuits(zl_ctx* ctx, const zl_g16_keys* k, const zl_circuit* const* circuits, const uint64_t* seeds, size_t count, zl_g16_proof* proofs) {
    if (!ctx || !k || (count && (!circuits || !seeds || !proofs))) return ZL_EINVAL;
    if (k->curve == ZL_BLS12_381) return prove_circuits_t<Bls12_381>(ctx, k->pc_bls, circuits, seeds, count, proofs);
    return prove_circuits_t<Bn254>(ctx, k->pc_bn, circuits, seeds, count, proofs);
}

// ---- ProvingContext / VerifyingKey wire formats (Groth16<E>::encode / decode / encode_verifying_key) ------------------------------
int zl_groth16_keys_to_bytes(const zl_g16_keys* k, uint8_t* out, size_t cap, size_t* len) {
    if (!k || !len || (!out && cap)) return ZL_EINVAL;
    Result<std::vector<uint8_t>> r = k->curve == ZL_BLS12_381 ? Groth16<Bls12_381>::encode(k->pc_bls, k->vk_bls) : Groth16<Bn254>::encode(k->pc_bn, k->vk_bn);
    if (!r.ok) return r.error.code;
    *len = r.value.size();
    if (cap < r.value.size()) return out ? ZL_EINVAL : ZL_OK;  // out == NULL, cap == 0: size query
    memcpy(out, r.value.data(), r.value.size());
    return ZL_OK;
}
int zl_groth16_keys_from_bytes(zl_ctx* ctx, zl_curve_t curve, const uint8_t* in, size_t len, unsigned flags, zl_g16_keys** out) {
    if (!ctx || !in || !out || (curve != ZL_BLS12_381 && curve != ZL_BN254) || (flags & ~ZL_CHECK)) return ZL_EINVAL;
    zl_g16_keys* k = new (std::nothrow) zl_g16_keys();
    if (!k) return ZL_ENOMEM;
    k->curve = curve;
    int rc = ZL_OK;
    if (curve == ZL_BLS12_381) {
        auto r = Groth16<Bls12_381>::decode(ctx, in, len, flags);
        if (!r.ok) rc = r.error.code; else { k->pc_bls = r.value.first; k->vk_bls = r.value.second; }
    } else {
        auto r = Groth16<Bn254>::decode(ctx, in, len, flags);
        if (!r.ok) rc = r.error.code; else { k->pc_bn = r.value.first; k->vk_bn = r.value.second; }
    }
    if (rc) { delete k; return rc; }
    *out = k;
    return ZL_OK;
}
// host-only validation of the same bytes (no ctx, no device): framing, lengths, canonical coordinates, flags; with ZL_CHECK the verifying-key points are also
// checked for curve / subgroup membership (the queries' curve checks happen on the device at upload).  What a service runs on untrusted input before it
// spends device memory on it, and the fuzz target of the sanitizer build (tests/test_sanitizers.py).
int zl_groth16_keys_parse(zl_curve_t curve, const uint8_t* in, size_t len, unsigned flags) {
    if (!in || (curve != ZL_BLS12_381 && curve != ZL_BN254) || (flags & ~ZL_CHECK)) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) {
        auto r = Groth16<Bls12_381>::decode(nullptr, in, len, flags);
        return r.ok ? ZL_OK : r.error.code;
    }
    auto r = Groth16<Bn254>::decode(nullptr, in, len, flags);
    return r.ok ? ZL_OK : r.error.code;
}
int zl_groth16_vk_to_bytes(const zl_g16_keys* k, uint8_t* out, size_t cap, size_t* len) {
    if (!k || !len || (!out && cap)) return ZL_EINVAL;
    const std::vector<uint8_t> v = k->curve == ZL_BLS12_381 ? Groth16<Bls12_381>::encode_verifying_key(k->vk_bls) : Groth16<Bn254>::encode_verifying_key(k->vk_bn);
    *len = v.size();
    if (cap < v.size()) return out ? ZL_EINVAL : ZL_OK;
    memcpy(out, v.data(), v.size());
    return ZL_OK;
}
size_t zl_point_bytes_uncompressed(zl_curve_t curve, zl_group_t group) { return 2 * zl_point_bytes(curve, group); }
int zl_point_to_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint8_t inf, uint8_t* out) {
    if (!xy || !out || !zl_point_bytes(curve, group)) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) { if (group == ZL_G1) serialize::BlsCodec::g1_to_uncompressed(xy, inf != 0, out); else serialize::BlsCodec::g2_to_uncompressed(xy, inf != 0, out); }
    else { if (group == ZL_G1) serialize::BnCodec::g1_to_uncompressed(xy, inf != 0, out); else serialize::BnCodec::g2_to_uncompressed(xy, inf != 0, out); }
    return ZL_OK;
}
int zl_point_from_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint8_t* in, int check, uint64_t* xy, uint8_t* inf) {
    if (!xy || !in || !inf || !zl_point_bytes(curve, group)) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) return group == ZL_G1 ? serialize::BlsCodec::g1_from_uncompressed(in, check != 0, xy, inf) : serialize::BlsCodec::g2_from_uncompressed(in, check != 0, xy, inf);
    return group == ZL_G1 ? serialize::BnCodec::g1_from_uncompressed(in, check != 0, xy, inf) : serialize::BnCodec::g2_from_uncompressed(in, check != 0, xy, inf);
}

// e(P, Q) after the final exponentiation: 12 canonical Fq coefficients of the w-polynomial (tests vs the oracle)
int zl_pairing(zl_curve_t curve, const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* out12) {
    if (!p_xy || !q_xy || !out12) return ZL_EINVAL;
    bool degenerate = false;  // a zero Miller value (inputs that are not points of the pairing groups): ZL_ENOTCURVE instead of a made-up number
    if (curve == ZL_BLS12_381) { pairing::BlsEngine::store(out12, pairing::BlsEngine::multi_pairing({p_xy}, {q_xy}, &degenerate)); return degenerate ? ZL_ENOTCURVE : ZL_OK; }
    if (curve == ZL_BN254) { pairing::BnEngine::store(out12, pairing::BnEngine::multi_pairing({p_xy}, {q_xy}, &degenerate)); return degenerate ? ZL_ENOTCURVE : ZL_OK; }
    return ZL_EINVAL;
}
// test hook (include/zl_backend_test.h): prod_i e(P_i, Q_i) through the lock-step Miller loops that Groth16::verify uses for its four pairings
int zl_test_pairing_product(zl_curve_t curve, size_t n, const uint64_t* ps_xy, const uint64_t* qs_xy, uint64_t* out12) {
    if (!out12 || ((!ps_xy || !qs_xy) && n) || n > 64) return ZL_EINVAL;
    if (curve != ZL_BLS12_381 && curve != ZL_BN254) return ZL_EINVAL;
    const size_t pw = curve == ZL_BLS12_381 ? 12 : 8;  // u64 words of a G1 point; a G2 point has twice as many
    std::vector<const uint64_t*> ps(n), qs(n);
    for (size_t i = 0; i < n; i++) { ps[i] = ps_xy + i * pw; qs[i] = qs_xy + i * 2 * pw; }
    bool degenerate = false;
    if (curve == ZL_BLS12_381) pairing::BlsEngine::store(out12, pairing::BlsEngine::multi_pairing(ps, qs, &degenerate));
    else pairing::BnEngine::store(out12, pairing::BnEngine::multi_pairing(ps, qs, &degenerate));
    return degenerate ? ZL_ENOTCURVE : ZL_OK;
}
// Groth16::verify: public_inputs = n x 4 u64 canonical (without the leading ONE); *ok = 1 / 0
int zl_groth16_verify(const zl_g16_keys* k, const uint64_t* public_inputs, size_t n, const zl_g16_proof* proof, int* ok) {
    if (!k || !proof || !ok || (!public_inputs && n)) return ZL_EINVAL;
    if (k->curve == ZL_BLS12_381) {
        std::vector<Fp<BLS12_381_Fr>> in(n);
        for (size_t i = 0; i < n; i++) memcpy(in[i].l, public_inputs + 4 * i, 32);
        auto r = Groth16<Bls12_381>::verify(k->vk_bls, in, *proof);
        if (!r.ok) return r.error.code;
        *ok = r.value ? 1 : 0;
    } else {
        std::vector<Fp<BN254_Fr>> in(n);
        for (size_t i = 0; i < n; i++) memcpy(in[i].l, public_inputs + 4 * i, 32);
        auto r = Groth16<Bn254>::verify(k->vk_bn, in, *proof);
        if (!r.ok) return r.error.code;
        *ok = r.value ? 1 : 0;
    }
    return ZL_OK;
}

// ---- wire formats (arkworks CanonicalSerialize, compressed; zl_serialize.h) -----------------------------------------------------
size_t zl_point_bytes(zl_curve_t curve, zl_group_t group) {
    if ((curve != ZL_BLS12_381 && curve != ZL_BN254) || (group != ZL_G1 && group != ZL_G2)) return 0;
    const size_t nb = curve == ZL_BLS12_381 ? 48 : 32;
    return group == ZL_G1 ? nb : 2 * nb;
}
int zl_point_to_bytes(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint8_t inf, uint8_t* out) {
    if (!xy || !out || !zl_point_bytes(curve, group)) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) { if (group == ZL_G1) serialize::BlsCodec::g1_to_bytes(xy, inf != 0, out); else serialize::BlsCodec::g2_to_bytes(xy, inf != 0, out); }
    else { if (group == ZL_G1) serialize::BnCodec::g1_to_bytes(xy, inf != 0, out); else serialize::BnCodec::g2_to_bytes(xy, inf != 0, out); }
    return ZL_OK;
}
int zl_point_from_bytes(zl_curve_t curve, zl_group_t group, const uint8_t* in, uint64_t* xy, uint8_t* inf) {
    if (!xy || !in || !inf || !zl_point_bytes(curve, group)) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) return group == ZL_G1 ? serialize::BlsCodec::g1_from_bytes(in, xy, inf) : serialize::BlsCodec::g2_from_bytes(in, xy, inf);
    return group == ZL_G1 ? serialize::BnCodec::g1_from_bytes(in, xy, inf) : serialize::BnCodec::g2_from_bytes(in, xy, inf);
}
size_t zl_groth16_proof_bytes(zl_curve_t curve) { return 2 * zl_point_bytes(curve, ZL_G1) + zl_point_bytes(curve, ZL_G2); }
int zl_groth16_proof_to_bytes(zl_curve_t curve, const zl_g16_proof* proof, uint8_t* out) {
    if (!proof || !out || !zl_groth16_proof_bytes(curve)) return ZL_EINVAL;
    const size_t n1 = zl_point_bytes(curve, ZL_G1), n2 = zl_point_bytes(curve, ZL_G2);
    int rc;
    if ((rc = zl_point_to_bytes(curve, ZL_G1, proof->a, proof->a_inf, out))) return rc;
    if ((rc = zl_point_to_bytes(curve, ZL_G2, proof->b, proof->b_inf, out + n1))) return rc;
    return zl_point_to_bytes(curve, ZL_G1, proof->c, proof->c_inf, out + n1 + n2);
}
int zl_groth16_proof_from_bytes(zl_curve_t curve, const uint8_t* in, size_t len, zl_g16_proof* proof) {
    if (!proof || !in || !zl_groth16_proof_bytes(curve) || len != zl_groth16_proof_bytes(curve)) return ZL_EINVAL;
    const size_t n1 = zl_point_bytes(curve, ZL_G1), n2 = zl_point_bytes(curve, ZL_G2);
    memset(proof, 0, sizeof *proof);
    int rc;
    if ((rc = zl_point_from_bytes(curve, ZL_G1, in, proof->a, &proof->a_inf))) return rc;
    if ((rc = zl_point_from_bytes(curve, ZL_G2, in + n1, proof->b, &proof->b_inf))) return rc;
    return zl_point_from_bytes(curve, ZL_G1, in + n1 + n2, proof->c, &proof->c_inf);
}

}  // extern "C"
