/* oracle/ec_tmpl.h -- short-Weierstrass (a = 0) Jacobian group + arkworks Pippenger template
 * (TEST INFRASTRUCTURE; CPU restatement used as checker and as the timed cpu_baseline "port").
 *
 * Restates ark-ec 0.3.0 (third-party, not vendored; pinned plugins/arkworks/Cargo.toml:120; surfaced by
 * `pub use ec;` plugins/arkworks/src/lib.rs:28-29):
 *   short_weierstrass_jacobian::GroupProjective::{add_assign_mixed (EFD madd-2007-bl),
 *   add_assign (add-2007-bl), double_in_place (dbl-2009-l), into_affine} and
 *   msm::VariableBaseMSM::multi_scalar_mul.
 * Include with: #define EC <prefix>  #define BF <coordinate-field prefix>  #define SC_BITS <Fr bits>
 */
#define E(name) CAT(EC, name)
#define K(name) CAT(BF, name)

typedef struct { K(t) x, y; int inf; } E(aff);
typedef struct { K(t) x, y, z; } E(jac);

static inline int E(jac_is_zero)(const E(jac) *p) { return K(is_zero)(&p->z); }
static inline void E(jac_set_zero)(E(jac) *p) {
    memset(p, 0, sizeof *p);
    p->x = K(ONE); /* ark: zero() = (1, 1, 0) */
    p->y = K(ONE);
}
static void E(jac_double)(E(jac) *p) {
    if (E(jac_is_zero)(p)) return;
    K(t) a, b, c, d, e, f, t;
    K(sqr)(&a, &p->x);
    K(sqr)(&b, &p->y);
    K(sqr)(&c, &b);
    K(add)(&d, &p->x, &b);
    K(sqr)(&d, &d);
    K(sub)(&d, &d, &a);
    K(sub)(&d, &d, &c);
    K(dbl)(&d, &d);
    K(dbl)(&e, &a);
    K(add)(&e, &e, &a);
    K(sqr)(&f, &e);
    K(mul)(&p->z, &p->z, &p->y);
    K(dbl)(&p->z, &p->z);
    K(sub)(&p->x, &f, &d);
    K(sub)(&p->x, &p->x, &d);
    K(sub)(&t, &d, &p->x);
    K(mul)(&t, &t, &e);
    K(dbl)(&c, &c);
    K(dbl)(&c, &c);
    K(dbl)(&c, &c);
    K(sub)(&p->y, &t, &c);
}
static void E(jac_add_mixed)(E(jac) *p, const E(aff) *q) {
    if (q->inf) return;
    if (E(jac_is_zero)(p)) {
        p->x = q->x;
        p->y = q->y;
        p->z = K(ONE);
        return;
    }
    K(t) z1z1, u2, s2, h, hh, i, j, r, v, t;
    K(sqr)(&z1z1, &p->z);
    K(mul)(&u2, &q->x, &z1z1);
    K(mul)(&s2, &q->y, &p->z);
    K(mul)(&s2, &s2, &z1z1);
    if (K(eq)(&p->x, &u2) && K(eq)(&p->y, &s2)) {
        E(jac_double)(p);
        return;
    }
    K(sub)(&h, &u2, &p->x);
    K(sqr)(&hh, &h);
    K(dbl)(&i, &hh);
    K(dbl)(&i, &i);
    K(mul)(&j, &h, &i);
    K(sub)(&r, &s2, &p->y);
    K(dbl)(&r, &r);
    K(mul)(&v, &p->x, &i);
    /* Z3 = (Z1+H)^2 - Z1Z1 - HH */
    K(add)(&t, &p->z, &h);
    K(sqr)(&t, &t);
    K(sub)(&t, &t, &z1z1);
    K(sub)(&p->z, &t, &hh);
    /* X3 = r^2 - J - 2V */
    K(sqr)(&p->x, &r);
    K(sub)(&p->x, &p->x, &j);
    K(sub)(&p->x, &p->x, &v);
    K(sub)(&p->x, &p->x, &v);
    /* Y3 = r (V - X3) - 2 Y1 J */
    K(mul)(&j, &p->y, &j);
    K(dbl)(&j, &j);
    K(sub)(&v, &v, &p->x);
    K(mul)(&v, &v, &r);
    K(sub)(&p->y, &v, &j);
}
static void E(jac_add)(E(jac) *p, const E(jac) *q) {
    if (E(jac_is_zero)(q)) return;
    if (E(jac_is_zero)(p)) { *p = *q; return; }
    K(t) z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    K(sqr)(&z1z1, &p->z);
    K(sqr)(&z2z2, &q->z);
    K(mul)(&u1, &p->x, &z2z2);
    K(mul)(&u2, &q->x, &z1z1);
    K(mul)(&s1, &p->y, &q->z);
    K(mul)(&s1, &s1, &z2z2);
    K(mul)(&s2, &q->y, &p->z);
    K(mul)(&s2, &s2, &z1z1);
    if (K(eq)(&u1, &u2) && K(eq)(&s1, &s2)) {
        E(jac_double)(p);
        return;
    }
    K(sub)(&h, &u2, &u1);
    K(dbl)(&i, &h);
    K(sqr)(&i, &i);
    K(mul)(&j, &h, &i);
    K(sub)(&r, &s2, &s1);
    K(dbl)(&r, &r);
    K(mul)(&v, &u1, &i);
    /* Z3 = ((Z1+Z2)^2 - Z1Z1 - Z2Z2) H */
    K(add)(&t, &p->z, &q->z);
    K(sqr)(&t, &t);
    K(sub)(&t, &t, &z1z1);
    K(sub)(&t, &t, &z2z2);
    K(mul)(&p->z, &t, &h);
    K(sqr)(&p->x, &r);
    K(sub)(&p->x, &p->x, &j);
    K(sub)(&p->x, &p->x, &v);
    K(sub)(&p->x, &p->x, &v);
    K(mul)(&s1, &s1, &j);
    K(dbl)(&s1, &s1);
    K(sub)(&v, &v, &p->x);
    K(mul)(&v, &v, &r);
    K(sub)(&p->y, &v, &s1);
}
static void E(jac_to_aff)(E(aff) *a, const E(jac) *p) {
    memset(a, 0, sizeof *a);
    if (E(jac_is_zero)(p)) { a->inf = 1; return; }
    K(t) zi, zi2;
    K(inv)(&zi, &p->z);
    K(sqr)(&zi2, &zi);
    K(mul)(&a->x, &p->x, &zi2);
    K(mul)(&zi2, &zi2, &zi);
    K(mul)(&a->y, &p->y, &zi2);
}
static void E(jac_from_aff)(E(jac) *p, const E(aff) *a) {
    if (a->inf) { E(jac_set_zero)(p); return; }
    p->x = a->x;
    p->y = a->y;
    p->z = K(ONE);
}
/* double-and-add scalar multiplication, scalar = 4 canonical limbs (definition-level, for tests) */
static void E(jac_mul)(E(jac) *r, const E(aff) *base, const uint64_t *k) {
    E(jac) acc;
    E(jac_set_zero)(&acc);
    for (int i = 255; i >= 0; i--) {
        E(jac_double)(&acc);
        if ((k[i / 64] >> (i % 64)) & 1) E(jac_add_mixed)(&acc, base);
    }
    *r = acc;
}

static inline uint64_t E(digit)(const uint64_t *s, int w_start, int c) {
    /* (s >> w_start) mod 2^c over a 256-bit little-endian integer */
    int limb = w_start / 64, sh = w_start % 64;
    uint64_t v = s[limb] >> sh;
    if (sh + c > 64 && limb + 1 < 4) v |= s[limb + 1] << (64 - sh);
    return v & ((1ull << c) - 1);
}
static inline int E(scalar_is)(const uint64_t *s, uint64_t v) {
    return s[0] == v && s[1] == 0 && s[2] == 0 && s[3] == 0;
}
static int E(ark_c)(size_t n) {
    if (n < 32) return 3;
    int cl = 0;
    while (((size_t)1 << cl) < n) cl++;
    return cl * 69 / 100 + 2;
}
/* one window of ark's VariableBaseMSM */
static void E(msm_window)(E(jac) *res, const E(aff) *bases, const uint64_t *scalars, size_t n, int w_start, int c,
                          E(jac) *buckets) {
    size_t nb = ((size_t)1 << c) - 1;
    E(jac_set_zero)(res);
    for (size_t b = 0; b < nb; b++) E(jac_set_zero)(&buckets[b]);
    for (size_t i = 0; i < n; i++) {
        const uint64_t *s = scalars + 4 * i;
        if (E(scalar_is)(s, 0)) continue;
        if (E(scalar_is)(s, 1)) {
            if (w_start == 0) E(jac_add_mixed)(res, &bases[i]);
            continue;
        }
        uint64_t d = E(digit)(s, w_start, c);
        if (d != 0) E(jac_add_mixed)(&buckets[d - 1], &bases[i]);
    }
    E(jac) run;
    E(jac_set_zero)(&run);
    for (size_t b = nb; b-- > 0;) {
        E(jac_add)(&run, &buckets[b]);
        E(jac_add)(res, &run);
    }
}
/* ark-ec 0.3.0 VariableBaseMSM::multi_scalar_mul.  threads <= 1: the reference's configuration (no
 * `parallel` feature, plugins/arkworks/Cargo.toml:25-110).  threads > 1: window-parallel, which is what
 * the arkworks `parallel` feature does (rayon over windows). */
static void E(msm_ark_c)(E(jac) *out, const E(aff) *bases, const uint64_t *scalars, size_t n, int threads, int c_override);
static void E(msm_ark)(E(jac) *out, const E(aff) *bases, const uint64_t *scalars, size_t n, int threads) {
    E(msm_ark_c)(out, bases, scalars, n, threads, 0);
}
/* c_override > 0: the window width ark's rule would pick for a LARGER input (timing a bounded sample of a big workload with the big
 * workload's window structure); results do not depend on c. */
static void E(msm_ark_c)(E(jac) *out, const E(aff) *bases, const uint64_t *scalars, size_t n, int threads, int c_override) {
    int c = c_override > 0 ? c_override : E(ark_c)(n);
    int nwin = (SC_BITS + c - 1) / c;
    E(jac) *ws = (E(jac) *)malloc(sizeof(E(jac)) * (size_t)nwin);
    size_t nb = ((size_t)1 << c) - 1;
#ifdef _OPENMP
    int nt = threads < 1 ? 1 : threads;
    if (nt > nwin) nt = nwin;
#pragma omp parallel num_threads(nt)
    {
        E(jac) *buckets = (E(jac) *)malloc(sizeof(E(jac)) * nb);
#pragma omp for schedule(dynamic, 1)
        for (int w = 0; w < nwin; w++) E(msm_window)(&ws[w], bases, scalars, n, w * c, c, buckets);
        free(buckets);
    }
#else
    (void)threads;
    E(jac) *buckets = (E(jac) *)malloc(sizeof(E(jac)) * nb);
    for (int w = 0; w < nwin; w++) E(msm_window)(&ws[w], bases, scalars, n, w * c, c, buckets);
    free(buckets);
#endif
    E(jac) total;
    E(jac_set_zero)(&total);
    for (int w = nwin - 1; w >= 1; w--) {
        E(jac_add)(&total, &ws[w]);
        for (int k = 0; k < c; k++) E(jac_double)(&total);
    }
    E(jac_add)(&total, &ws[0]);
    *out = total;
    free(ws);
}
/* All-core MSM as a (chunk x window) grid (NOT what arkworks 0.3.0 does -- its `parallel` feature stops at one thread per window): the
 * input is cut into contiguous chunks so that chunks x windows ~ 4 x threads, every (chunk, window) pair is one task running ark's window
 * routine (window width by ark's rule for the chunk length; bucket arrays of a chunk-sized problem stay cache-sized), then a Horner per
 * chunk and the chunk results are added.  It uses every core on the same algorithm; it is NOT always the fastest arrangement (on a 256-thread box the window-parallel run of a 2^22 prefix on 15 threads measured faster per point: bench.py reports both and takes the better). */
static void E(msm_chunked)(E(jac) *out, const E(aff) *bases, const uint64_t *scalars, size_t n, int threads) {
    int nt = threads < 1 ? 1 : threads;
    size_t chunks = 1;
    int c = E(ark_c)(n), nwin = (SC_BITS + c - 1) / c;
    /* >= 4 tasks per thread (dynamic schedule): with tasks ~ threads the last round of a 272-task grid ran on 16 of 256 threads */
    for (chunks = 1; chunks < 4 * (size_t)nt && chunks * 4096 < n; chunks++) {
        c = E(ark_c)((n + chunks - 1) / chunks);
        nwin = (SC_BITS + c - 1) / c;
        if (chunks * (size_t)nwin >= 4 * (size_t)nt) break;
    }
    if (chunks < 1) chunks = 1;
    size_t per = (n + chunks - 1) / chunks;
    c = E(ark_c)(per ? per : 1);
    nwin = (SC_BITS + c - 1) / c;
    size_t tasks = chunks * (size_t)nwin, nb = ((size_t)1 << c) - 1;
    E(jac) *ws = (E(jac) *)malloc(sizeof(E(jac)) * tasks);
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        E(jac) *buckets = (E(jac) *)malloc(sizeof(E(jac)) * nb);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (size_t t = 0; t < tasks; t++) {
            size_t ch = t / (size_t)nwin, w = t % (size_t)nwin;
            size_t lo = ch * per, hi = lo + per > n ? n : lo + per;
            if (lo >= hi) { E(jac_set_zero)(&ws[t]); continue; }
            E(msm_window)(&ws[t], bases + lo, scalars + 4 * lo, hi - lo, (int)w * c, c, buckets);
        }
        free(buckets);
    }
    E(jac) total;
    E(jac_set_zero)(&total);
    for (size_t ch = 0; ch < chunks; ch++) {
        E(jac) acc;
        E(jac_set_zero)(&acc);
        for (int w = nwin - 1; w >= 1; w--) {
            E(jac_add)(&acc, &ws[ch * (size_t)nwin + (size_t)w]);
            for (int k = 0; k < c; k++) E(jac_double)(&acc);
        }
        E(jac_add)(&acc, &ws[ch * (size_t)nwin]);
        E(jac_add)(&total, &acc);
    }
    *out = total;
    free(ws);
}
/* definition-level MSM: sum of double-and-add scalar multiples */
static void E(msm_naive)(E(jac) *out, const E(aff) *bases, const uint64_t *scalars, size_t n) {
    E(jac) acc, t;
    E(jac_set_zero)(&acc);
    for (size_t i = 0; i < n; i++) {
        E(jac_mul)(&t, &bases[i], scalars + 4 * i);
        E(jac_add)(&acc, &t);
    }
    *out = acc;
}
#undef E
#undef K
