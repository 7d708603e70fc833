/* oracle/ntt_tmpl.h -- radix-2 evaluation-domain template (TEST INFRASTRUCTURE).
 * Restates ark-poly 0.3.0 Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
 * (third-party, not vendored; pinned plugins/arkworks/Cargo.toml:139; surfaced by `pub use poly;`
 * plugins/arkworks/src/lib.rs:70-71; reached from groth16.rs:454 via R1CStoQAP::witness_map):
 * natural order in and out; group_gen = TWO_ADIC_ROOT^(2^(TWO_ADICITY-log_n)); inverse scales by
 * n^-1; coset variants multiply coefficient i by g^i before the forward transform / by g^-i after the
 * inverse, g = multiplicative generator.  Roots are tabulated (n/2 entries) as upstream does.
 * Include with: #define NT <prefix> #define FP <Fr prefix>; NT##_ROOT_CANON, NT##_GEN, NT##_TWO_ADICITY by includer
 */
#define N_(name) CAT(NT, name)
#define R(name) CAT(FP, name)

static void N_(domain_root)(R(t) *w, unsigned log_n) {
    R(from_canon)(w, N_(ROOT_CANON));
    for (unsigned i = log_n; i < N_(TWO_ADICITY); i++) R(sqr)(w, w);
}
static void N_(bitrev)(R(t) *a, size_t n) {
    size_t j = 0;
    for (size_t i = 1; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { R(t) t = a[i]; a[i] = a[j]; a[j] = t; }
    }
}
static void N_(distribute_powers)(R(t) *a, size_t n, const R(t) *g) {
    R(t) pw = R(ONE);
    for (size_t i = 0; i < n; i++) {
        R(mul)(&a[i], &a[i], &pw);
        R(mul)(&pw, &pw, g);
    }
}
static void N_(pow_u64)(R(t) *r, const R(t) *b, uint64_t e) {
    R(t) acc = R(ONE), x = *b;
    while (e) {
        if (e & 1) R(mul)(&acc, &acc, &x);
        R(sqr)(&x, &x);
        e >>= 1;
    }
    *r = acc;
}
/* Multi-threaded arrangement of the SAME in-order radix-2 transform for bench.py's all-core CPU baseline (arkworks' `parallel`
 * feature also parallelises its FFT; the reference builds without it): root table / coset powers by chunks, butterflies of a stage
 * split evenly over the threads.  Bit-identical results to N_(transform). */
static void N_(transform_mt)(R(t) *a, unsigned log_n, int inverse, int coset, int threads) {
    size_t n = (size_t)1 << log_n;
    int nt = threads < 1 ? 1 : threads;
    R(t) w, g;
    N_(domain_root)(&w, log_n);
    R(from_u64)(&g, N_(GEN));
    if (inverse) R(inv)(&w, &w);
    const size_t CH = 4096;
    if (coset && !inverse) {
#pragma omp parallel for num_threads(nt) schedule(static)
        for (size_t c0 = 0; c0 < n; c0 += CH) {
            R(t) pw;
            N_(pow_u64)(&pw, &g, (uint64_t)c0);
            for (size_t i = c0; i < c0 + CH && i < n; i++) { R(mul)(&a[i], &a[i], &pw); R(mul)(&pw, &pw, &g); }
        }
    }
    if (n > 1) {
        R(t) *roots = (R(t) *)malloc(sizeof(R(t)) * (n / 2));
#pragma omp parallel for num_threads(nt) schedule(static)
        for (size_t c0 = 0; c0 < n / 2; c0 += CH) {
            R(t) pw;
            N_(pow_u64)(&pw, &w, (uint64_t)c0);
            for (size_t i = c0; i < c0 + CH && i < n / 2; i++) { roots[i] = pw; R(mul)(&pw, &pw, &w); }
        }
#pragma omp parallel for num_threads(nt) schedule(static)
        for (size_t i = 0; i < n; i++) {  /* bit reversal: every pair is swapped by its smaller index */
            size_t j = 0;
            for (unsigned b = 0; b < log_n; b++) j |= ((i >> b) & 1) << (log_n - 1 - b);
            if (i < j) { R(t) t = a[i]; a[i] = a[j]; a[j] = t; }
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            size_t half = len / 2, step = n / len;
#pragma omp parallel for num_threads(nt) schedule(static)
            for (size_t j = 0; j < n / 2; j++) {
                size_t k = j & (half - 1), i = (j - k) * 2;
                R(t) u = a[i + k], v;
                R(mul)(&v, &a[i + k + half], &roots[k * step]);
                R(add)(&a[i + k], &u, &v);
                R(sub)(&a[i + k + half], &u, &v);
            }
        }
        free(roots);
    }
    if (inverse) {
        R(t) ninv;
        R(from_u64)(&ninv, (uint64_t)n);
        R(inv)(&ninv, &ninv);
        R(t) gi;
        R(inv)(&gi, &g);
#pragma omp parallel for num_threads(nt) schedule(static)
        for (size_t c0 = 0; c0 < n; c0 += CH) {
            R(t) pw;
            if (coset) N_(pow_u64)(&pw, &gi, (uint64_t)c0);
            for (size_t i = c0; i < c0 + CH && i < n; i++) {
                R(mul)(&a[i], &a[i], &ninv);
                if (coset) { R(mul)(&a[i], &a[i], &pw); R(mul)(&pw, &pw, &gi); }
            }
        }
    }
}
/* a: n = 2^log_n Montgomery-form elements, transformed in place */
static void N_(transform)(R(t) *a, unsigned log_n, int inverse, int coset) {
    size_t n = (size_t)1 << log_n;
    R(t) w, g;
    N_(domain_root)(&w, log_n);
    R(from_u64)(&g, N_(GEN));
    if (inverse) R(inv)(&w, &w);
    if (coset && !inverse) N_(distribute_powers)(a, n, &g);
    if (n > 1) {
        R(t) *roots = (R(t) *)malloc(sizeof(R(t)) * (n / 2));
        roots[0] = R(ONE);
        for (size_t i = 1; i < n / 2; i++) R(mul)(&roots[i], &roots[i - 1], &w);
        N_(bitrev)(a, n);
        for (size_t len = 2; len <= n; len <<= 1) {
            size_t half = len / 2, step = n / len;
            for (size_t i = 0; i < n; i += len) {
                for (size_t k = 0; k < half; k++) {
                    R(t) u = a[i + k], v;
                    R(mul)(&v, &a[i + k + half], &roots[k * step]);
                    R(add)(&a[i + k], &u, &v);
                    R(sub)(&a[i + k + half], &u, &v);
                }
            }
        }
        free(roots);
    }
    if (inverse) {
        R(t) ninv;
        R(from_u64)(&ninv, (uint64_t)n);
        R(inv)(&ninv, &ninv);
        for (size_t i = 0; i < n; i++) R(mul)(&a[i], &a[i], &ninv);
        if (coset) {
            R(t) gi;
            R(inv)(&gi, &g);
            N_(distribute_powers)(a, n, &gi);
        }
    }
}
#undef N_
#undef R
