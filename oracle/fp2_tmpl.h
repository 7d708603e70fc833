/* oracle/fp2_tmpl.h -- Fq2 = Fq[u]/(u^2+1) template (TEST INFRASTRUCTURE).
 * Restates ark-ff 0.3.0 QuadExtField with NONRESIDUE = -1 (both BLS12-381 and BN254 Fq2), the
 * coordinate field of G2 used by the 5th MSM of ark-groth16's create_proof (b_g2_query;
 * reference call site plugins/arkworks/src/groth16.rs:454).
 * Include with: #define FP2 <prefix>  #define FP <base prefix>
 */
#define G(name) CAT(FP2, name)
#define B(name) CAT(FP, name)

typedef struct { B(t) c0, c1; } G(t);
static G(t) G(ONE);

static inline int G(is_zero)(const G(t) *a) { return B(is_zero)(&a->c0) && B(is_zero)(&a->c1); }
static inline int G(eq)(const G(t) *a, const G(t) *b) { return B(eq)(&a->c0, &b->c0) && B(eq)(&a->c1, &b->c1); }
static inline void G(add)(G(t) *r, const G(t) *a, const G(t) *b) { B(add)(&r->c0, &a->c0, &b->c0); B(add)(&r->c1, &a->c1, &b->c1); }
static inline void G(sub)(G(t) *r, const G(t) *a, const G(t) *b) { B(sub)(&r->c0, &a->c0, &b->c0); B(sub)(&r->c1, &a->c1, &b->c1); }
static inline void G(dbl)(G(t) *r, const G(t) *a) { G(add)(r, a, a); }
static inline void G(neg)(G(t) *r, const G(t) *a) { B(neg)(&r->c0, &a->c0); B(neg)(&r->c1, &a->c1); }
static inline void G(mul)(G(t) *r, const G(t) *a, const G(t) *b) {
    B(t) v0, v1, s, t, u;
    B(mul)(&v0, &a->c0, &b->c0);
    B(mul)(&v1, &a->c1, &b->c1);
    B(add)(&s, &a->c0, &a->c1);
    B(add)(&t, &b->c0, &b->c1);
    B(mul)(&u, &s, &t);
    B(sub)(&u, &u, &v0);
    B(sub)(&r->c1, &u, &v1);
    B(sub)(&r->c0, &v0, &v1);
}
static inline void G(sqr)(G(t) *r, const G(t) *a) { G(mul)(r, a, a); }
static void G(inv)(G(t) *r, const G(t) *a) {
    B(t) n, t, ni;
    B(sqr)(&n, &a->c0);
    B(sqr)(&t, &a->c1);
    B(add)(&n, &n, &t);
    B(inv)(&ni, &n);
    B(mul)(&r->c0, &a->c0, &ni);
    B(mul)(&t, &a->c1, &ni);
    B(neg)(&r->c1, &t);
}
static void G(init)(void) {
    G(ONE).c0 = B(ONE);
    memset(&G(ONE).c1, 0, sizeof(B(t)));
}
#undef G
#undef B
