/* oracle/fp_tmpl.h -- prime-field template (TEST INFRASTRUCTURE; CPU restatement, not product code).
 *
 * Restates ark-ff 0.3.0 Fp256/Fp384 (third-party crate, not vendored under /root/reference; pinned at
 * plugins/arkworks/Cargo.toml:129; surfaced by `pub use ff::*`, plugins/arkworks/src/ff.rs:6):
 * NL little-endian 64-bit limbs holding value*R mod p, R = 2^(64*NL); multiplication is Montgomery
 * CIOS; every result is fully reduced to [0,p), so limbs are unique and byte-comparable.
 *
 * Include with:  #define FP <prefix>   #define NL <limbs>   (FP##_MOD / FP##_INV provided by includer)
 */
#ifndef CAT
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#endif
#define F(name) CAT(FP, name)

typedef struct { uint64_t l[NL]; } F(t);

static F(t) F(R2);  /* R^2 mod p, filled by F(init) */
static F(t) F(ONE); /* R mod p */

static inline int F(is_zero)(const F(t) *a) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i];
    return acc == 0;
}
static inline int F(eq)(const F(t) *a, const F(t) *b) {
    uint64_t acc = 0;
    for (int i = 0; i < NL; i++) acc |= a->l[i] ^ b->l[i];
    return acc == 0;
}
/* a >= modulus ? */
static inline int F(geq_mod)(const uint64_t *a) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > F(MOD)[i]) return 1;
        if (a[i] < F(MOD)[i]) return 0;
    }
    return 1;
}
static inline void F(sub_mod_raw)(uint64_t *a) {
    unsigned __int128 br = 0;
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a[i] - F(MOD)[i] - (uint64_t)br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline void F(add)(F(t) *r, const F(t) *a, const F(t) *b) {
    unsigned __int128 c = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) {
        c += (unsigned __int128)a->l[i] + b->l[i];
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    /* moduli used here leave at least one spare top bit, so no carry out of limb NL-1 */
    if (F(geq_mod)(t)) F(sub_mod_raw)(t);
    memcpy(r->l, t, sizeof t);
}
static inline void F(dbl)(F(t) *r, const F(t) *a) { F(add)(r, a, a); }
static inline void F(sub)(F(t) *r, const F(t) *a, const F(t) *b) {
    unsigned __int128 br = 0;
    uint64_t t[NL];
    for (int i = 0; i < NL; i++) {
        unsigned __int128 d = (unsigned __int128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        unsigned __int128 c = 0;
        for (int i = 0; i < NL; i++) {
            c += (unsigned __int128)t[i] + F(MOD)[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}
static inline void F(neg)(F(t) *r, const F(t) *a) {
    F(t) z;
    memset(&z, 0, sizeof z);
    if (F(is_zero)(a)) { *r = z; return; }
    F(sub)(r, &z, a);
}
/* Montgomery CIOS: r = a*b*R^-1 mod p */
static inline void F(mul)(F(t) *r, const F(t) *a, const F(t) *b) {
    uint64_t t[NL + 2];
    memset(t, 0, sizeof t);
    for (int i = 0; i < NL; i++) {
        unsigned __int128 c = 0;
        for (int j = 0; j < NL; j++) {
            c += (unsigned __int128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL] = (uint64_t)c;
        t[NL + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F(INV);
        c = (unsigned __int128)m * F(MOD)[0] + t[0];
        c >>= 64;
        for (int j = 1; j < NL; j++) {
            c += (unsigned __int128)m * F(MOD)[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[NL];
        t[NL - 1] = (uint64_t)c;
        t[NL] = t[NL + 1] + (uint64_t)(c >> 64);
    }
    if (t[NL] || F(geq_mod)(t)) F(sub_mod_raw)(t);
    memcpy(r->l, t, NL * sizeof(uint64_t));
}
static inline void F(sqr)(F(t) *r, const F(t) *a) { F(mul)(r, a, a); }

static inline void F(from_canon)(F(t) *r, const uint64_t *canon) {
    F(t) c;
    memcpy(c.l, canon, sizeof c.l);
    F(mul)(r, &c, &F(R2));
}
static inline void F(to_canon)(uint64_t *canon, const F(t) *a) {
    F(t) one, out;
    memset(&one, 0, sizeof one);
    one.l[0] = 1;
    F(mul)(&out, a, &one);
    memcpy(canon, out.l, sizeof out.l);
}
static inline void F(from_u64)(F(t) *r, uint64_t v) {
    uint64_t c[NL];
    memset(c, 0, sizeof c);
    c[0] = v;
    F(from_canon)(r, c);
}
/* r = a^e, e given as NL canonical limbs (square-and-multiply, MSB first) */
static void F(pow)(F(t) *r, const F(t) *a, const uint64_t *e, int elimbs) {
    F(t) acc = F(ONE);
    for (int i = elimbs * 64 - 1; i >= 0; i--) {
        F(sqr)(&acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) F(mul)(&acc, &acc, a);
    }
    *r = acc;
}
/* Fermat inversion a^(p-2); 0 maps to 0 (callers test is_zero first, like ark's Option) */
static void F(inv)(F(t) *r, const F(t) *a) {
    uint64_t e[NL];
    memcpy(e, F(MOD), sizeof e);
    e[0] -= 2; /* all moduli here are odd and > 2, low limb >= 3: no borrow */
    F(pow)(r, a, e, NL);
}
static void F(init)(void) {
    /* R mod p and R^2 mod p by repeated doubling of 1 in the plain (non-Montgomery) domain */
    F(t) x;
    memset(&x, 0, sizeof x);
    x.l[0] = 1;
    for (int i = 0; i < 64 * NL; i++) F(add)(&x, &x, &x);
    F(ONE) = x;
    for (int i = 0; i < 64 * NL; i++) F(add)(&x, &x, &x);
    F(R2) = x;
}
#undef F
