/* oracle/zl_oracle.c -- CPU oracle: instantiations + C API.  TEST INFRASTRUCTURE ONLY (see zl_oracle.h).
 *
 * The arithmetic restated here lives in arkworks 0.3.x crates that are not vendored under
 * /root/reference (plugins/arkworks/Cargo.toml:113-146); the reference reaches it at
 * plugins/arkworks/src/groth16.rs:438 (setup) and :454 (prove).  Each template header cites the
 * upstream item it follows.  Parity: see zl_oracle.h (unpinned at MSM/NTT; Fr pinned by fixtures).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "zl_oracle.h"
#include "zl_consts.h"

/* ---- fields ---- */
#define FP blsq
#define NL 6
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP blsr
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bnq
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
#define FP bnr
#define NL 4
#include "fp_tmpl.h"
#undef FP
#undef NL
/* ---- Fq2 ---- */
#define FP2 blsq2
#define FP blsq
#include "fp2_tmpl.h"
#undef FP2
#undef FP
#define FP2 bnq2
#define FP bnq
#include "fp2_tmpl.h"
#undef FP2
#undef FP
/* ---- groups ---- */
#define EC blsg1
#define BF blsq
#define SC_BITS 255
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC blsg2
#define BF blsq2
#include "ec_tmpl.h"
#undef EC
#undef BF
#undef SC_BITS
#define EC bng1
#define BF bnq
#define SC_BITS 254
#include "ec_tmpl.h"
#undef EC
#undef BF
#define EC bng2
#define BF bnq2
#include "ec_tmpl.h"
#undef EC
#undef BF
#undef SC_BITS
/* ---- domains ---- */
#define NT blsntt
#define FP blsr
#include "ntt_tmpl.h"
#undef NT
#undef FP
#define NT bnntt
#define FP bnr
#include "ntt_tmpl.h"
#undef NT
#undef FP

__attribute__((constructor)) static void zlo_init(void) {
    blsq_init();
    blsr_init();
    bnq_init();
    bnr_init();
    blsq2_init();
    bnq2_init();
}

int zlo_threads_available(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- field API */
#define FIELD_SWITCH(fid, M)      \
    switch (fid) {                \
    case ZLO_F_BLS_FQ: M(blsq, 6) \
    case ZLO_F_BLS_FR: M(blsr, 4) \
    case ZLO_F_BN_FQ: M(bnq, 4)   \
    case ZLO_F_BN_FR: M(bnr, 4)   \
    default: return -1;           \
    }

int zlo_field_op(int fid, int op, const uint64_t *a, const uint64_t *b, uint64_t *r) {
#define M(P, N)                                           \
    {                                                     \
        P##_t x, y, z;                                    \
        P##_from_canon(&x, a);                            \
        if (op != 3) P##_from_canon(&y, b);               \
        if (op == 0) P##_mul(&z, &x, &y);                 \
        else if (op == 1) P##_add(&z, &x, &y);            \
        else if (op == 2) P##_sub(&z, &x, &y);            \
        else if (op == 3) P##_inv(&z, &x);                \
        else return -1;                                   \
        P##_to_canon(r, &z);                              \
        return 0;                                         \
    }
    FIELD_SWITCH(fid, M)
#undef M
}
int zlo_field_to_mont(int fid, const uint64_t *in, uint64_t *out, size_t n) {
#define M(P, N)                                                                  \
    {                                                                            \
        for (size_t i = 0; i < n; i++) {                                         \
            P##_t x;                                                             \
            P##_from_canon(&x, in + (size_t)N * i);                              \
            memcpy(out + (size_t)N * i, x.l, sizeof x.l);                        \
        }                                                                        \
        return 0;                                                                \
    }
    FIELD_SWITCH(fid, M)
#undef M
}
int zlo_field_from_mont(int fid, const uint64_t *in, uint64_t *out, size_t n) {
#define M(P, N)                                                                  \
    {                                                                            \
        for (size_t i = 0; i < n; i++) {                                         \
            P##_t x;                                                             \
            memcpy(x.l, in + (size_t)N * i, sizeof x.l);                         \
            P##_to_canon(out + (size_t)N * i, &x);                               \
        }                                                                        \
        return 0;                                                                \
    }
    FIELD_SWITCH(fid, M)
#undef M
}

/* ---------------------------------------------------------------- G1 API */
#define DEF_G1(EC, FQ, NLQ)                                                                                       \
    static void EC##_load_aff(EC##_aff *a, const uint64_t *xy, int mont) {                                        \
        uint64_t acc = 0;                                                                                         \
        for (int i = 0; i < 2 * NLQ; i++) acc |= xy[i];                                                           \
        memset(a, 0, sizeof *a);                                                                                  \
        if (!acc) { a->inf = 1; return; }                                                                         \
        if (mont) {                                                                                               \
            memcpy(a->x.l, xy, sizeof a->x.l);                                                                    \
            memcpy(a->y.l, xy + NLQ, sizeof a->y.l);                                                              \
        } else {                                                                                                  \
            FQ##_from_canon(&a->x, xy);                                                                           \
            FQ##_from_canon(&a->y, xy + NLQ);                                                                     \
        }                                                                                                         \
    }                                                                                                             \
    static void EC##_store_aff(uint64_t *xy, uint8_t *inf, const EC##_jac *p) {                                   \
        EC##_aff a;                                                                                               \
        EC##_jac_to_aff(&a, p);                                                                                   \
        memset(xy, 0, sizeof(uint64_t) * 2 * NLQ);                                                                \
        if (inf) *inf = (uint8_t)a.inf;                                                                           \
        if (a.inf) return;                                                                                        \
        FQ##_to_canon(xy, &a.x);                                                                                  \
        FQ##_to_canon(xy + NLQ, &a.y);                                                                            \
    }                                                                                                             \
    static int EC##_msm_api(const uint64_t *bases, int mont, const uint64_t *scalars, size_t n, int algo,         \
                            int threads, uint64_t *out_xy, uint8_t *out_inf) {                                    \
        EC##_aff *b = (EC##_aff *)malloc(sizeof(EC##_aff) * (n ? n : 1));                                        \
        if (!b) return -2;                                                                                        \
        for (size_t i = 0; i < n; i++) EC##_load_aff(&b[i], bases + (size_t)2 * NLQ * i, mont);                   \
        EC##_jac r;                                                                                               \
        if (algo == 0) EC##_msm_ark(&r, b, scalars, n, threads);                                                  \
        else EC##_msm_naive(&r, b, scalars, n);                                                                   \
        EC##_store_aff(out_xy, out_inf, &r);                                                                      \
        free(b);                                                                                                  \
        return 0;                                                                                                 \
    }                                                                                                             \
    /* timed variant for bench.py's cpu_baseline: parallel (untimed) load of the bases, *seconds = the MSM alone. */ \
    static int EC##_msm_ex_api(const uint64_t *bases, int mont, const uint64_t *scalars, size_t n, int algo, int c_override, \
                               int threads, uint64_t *out_xy, uint8_t *out_inf, double *seconds) {                 \
        EC##_aff *b = (EC##_aff *)malloc(sizeof(EC##_aff) * (n ? n : 1));                                        \
        if (!b) return -2;                                                                                        \
        _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++)                               \
            EC##_load_aff(&b[i], bases + (size_t)2 * NLQ * i, mont);                                              \
        EC##_jac r;                                                                                               \
        const double t0 = omp_get_wtime();                                                                        \
        if (algo == 0) EC##_msm_ark_c(&r, b, scalars, n, threads, c_override);                                    \
        else if (algo == 2) EC##_msm_chunked(&r, b, scalars, n, threads);                                         \
        else EC##_msm_naive(&r, b, scalars, n);                                                                   \
        if (seconds) *seconds = omp_get_wtime() - t0;                                                             \
        EC##_store_aff(out_xy, out_inf, &r);                                                                      \
        free(b);                                                                                                  \
        return 0;                                                                                                 \
    }                                                                                                             \
    static int EC##_mul_gen_api(const uint64_t *gen, const uint64_t *k, size_t n, uint64_t *out_xy) {             \
        EC##_aff g;                                                                                               \
        EC##_load_aff(&g, gen, 0);                                                                                \
        /* fixed-base 8-bit windowed table: T[w][d] = d * 2^(8w) * G */                                           \
        EC##_aff *tab = (EC##_aff *)malloc(sizeof(EC##_aff) * 32 * 256);                                          \
        if (!tab) return -2;                                                                                      \
        EC##_jac base;                                                                                            \
        EC##_jac_from_aff(&base, &g);                                                                             \
        for (int w = 0; w < 32; w++) {                                                                            \
            EC##_jac acc;                                                                                         \
            EC##_jac_set_zero(&acc);                                                                              \
            tab[w * 256].inf = 1;                                                                                 \
            for (int d = 1; d < 256; d++) {                                                                       \
                EC##_jac_add(&acc, &base);                                                                        \
                EC##_jac_to_aff(&tab[w * 256 + d], &acc);                                                         \
            }                                                                                                     \
            for (int s = 0; s < 8; s++) EC##_jac_double(&base);                                                   \
        }                                                                                                         \
        _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++) {                             \
            EC##_jac acc;                                                                                         \
            EC##_jac_set_zero(&acc);                                                                              \
            for (int w = 0; w < 32; w++) {                                                                        \
                unsigned d = (unsigned)(k[4 * i + w / 8] >> (8 * (w % 8))) & 0xff;                                \
                if (d) EC##_jac_add_mixed(&acc, &tab[w * 256 + d]);                                               \
            }                                                                                                     \
            EC##_store_aff(out_xy + (size_t)2 * NLQ * i, NULL, &acc);                                             \
        }                                                                                                         \
        free(tab);                                                                                                \
        return 0;                                                                                                 \
    }

#define NLQ1_bls 6
DEF_G1(blsg1, blsq, 6)
DEF_G1(bng1, bnq, 4)

int zlo_msm_g1(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo,
               int threads, uint64_t *out_xy, uint8_t *out_inf) {
    if (curve == ZLO_BLS12_381) return blsg1_msm_api(bases, bases_mont, scalars, n, algo, threads, out_xy, out_inf);
    if (curve == ZLO_BN254) return bng1_msm_api(bases, bases_mont, scalars, n, algo, threads, out_xy, out_inf);
    return -1;
}
int zlo_msm_g1_ex(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo, int c_override,
                  int threads, uint64_t *out_xy, uint8_t *out_inf, double *seconds) {
    if (curve == ZLO_BLS12_381) return blsg1_msm_ex_api(bases, bases_mont, scalars, n, algo, c_override, threads, out_xy, out_inf, seconds);
    if (curve == ZLO_BN254) return bng1_msm_ex_api(bases, bases_mont, scalars, n, algo, c_override, threads, out_xy, out_inf, seconds);
    return -1;
}
int zlo_g1_mul_gen(int curve, const uint64_t *k, size_t n, uint64_t *out_xy) {
    if (curve == ZLO_BLS12_381) return blsg1_mul_gen_api(bls_G1_GEN, k, n, out_xy);
    if (curve == ZLO_BN254) return bng1_mul_gen_api(bn_G1_GEN, k, n, out_xy);
    return -1;
}
int zlo_g1_mul(int curve, const uint64_t *p_xy, const uint64_t *k, uint64_t *out_xy, uint8_t *out_inf) {
    if (curve == ZLO_BLS12_381) {
        blsg1_aff a;
        blsg1_jac r;
        blsg1_load_aff(&a, p_xy, 0);
        blsg1_jac_mul(&r, &a, k);
        blsg1_store_aff(out_xy, out_inf, &r);
        return 0;
    }
    if (curve == ZLO_BN254) {
        bng1_aff a;
        bng1_jac r;
        bng1_load_aff(&a, p_xy, 0);
        bng1_jac_mul(&r, &a, k);
        bng1_store_aff(out_xy, out_inf, &r);
        return 0;
    }
    return -1;
}

/* ---------------------------------------------------------------- G2 API */
#define DEF_G2(EC, FQ2, FQ, NLQ)                                                                                  \
    static void EC##_load_aff(EC##_aff *a, const uint64_t *xy, int mont) {                                        \
        uint64_t acc = 0;                                                                                         \
        for (int i = 0; i < 4 * NLQ; i++) acc |= xy[i];                                                           \
        memset(a, 0, sizeof *a);                                                                                  \
        if (!acc) { a->inf = 1; return; }                                                                         \
        FQ##_t *dst[4] = {&a->x.c0, &a->x.c1, &a->y.c0, &a->y.c1};                                                \
        for (int k = 0; k < 4; k++) {                                                                             \
            if (mont) memcpy(dst[k]->l, xy + k * NLQ, sizeof dst[k]->l);                                          \
            else FQ##_from_canon(dst[k], xy + k * NLQ);                                                           \
        }                                                                                                         \
    }                                                                                                             \
    static void EC##_store_aff(uint64_t *xy, uint8_t *inf, const EC##_jac *p) {                                   \
        EC##_aff a;                                                                                               \
        EC##_jac_to_aff(&a, p);                                                                                   \
        memset(xy, 0, sizeof(uint64_t) * 4 * NLQ);                                                                \
        if (inf) *inf = (uint8_t)a.inf;                                                                           \
        if (a.inf) return;                                                                                        \
        FQ##_to_canon(xy, &a.x.c0);                                                                               \
        FQ##_to_canon(xy + NLQ, &a.x.c1);                                                                         \
        FQ##_to_canon(xy + 2 * NLQ, &a.y.c0);                                                                     \
        FQ##_to_canon(xy + 3 * NLQ, &a.y.c1);                                                                     \
    }                                                                                                             \
    static int EC##_msm_api(const uint64_t *bases, int mont, const uint64_t *scalars, size_t n, int algo,         \
                            int threads, uint64_t *out_xy, uint8_t *out_inf) {                                    \
        EC##_aff *b = (EC##_aff *)malloc(sizeof(EC##_aff) * (n ? n : 1));                                        \
        if (!b) return -2;                                                                                        \
        for (size_t i = 0; i < n; i++) EC##_load_aff(&b[i], bases + (size_t)4 * NLQ * i, mont);                   \
        EC##_jac r;                                                                                               \
        if (algo == 0) EC##_msm_ark(&r, b, scalars, n, threads);                                                  \
        else EC##_msm_naive(&r, b, scalars, n);                                                                   \
        EC##_store_aff(out_xy, out_inf, &r);                                                                      \
        free(b);                                                                                                  \
        return 0;                                                                                                 \
    }                                                                                                             \
    static int EC##_mul_gen_api(const uint64_t *gen, const uint64_t *k, size_t n, uint64_t *out_xy) {             \
        EC##_aff g;                                                                                               \
        EC##_load_aff(&g, gen, 0);                                                                                \
        _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++) {                             \
            EC##_jac r;                                                                                           \
            EC##_jac_mul(&r, &g, k + 4 * i);                                                                      \
            EC##_store_aff(out_xy + (size_t)4 * NLQ * i, NULL, &r);                                               \
        }                                                                                                         \
        return 0;                                                                                                 \
    }
DEF_G2(blsg2, blsq2, blsq, 6)
DEF_G2(bng2, bnq2, bnq, 4)

int zlo_msm_g2(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo,
               int threads, uint64_t *out_xy, uint8_t *out_inf) {
    if (curve == ZLO_BLS12_381) return blsg2_msm_api(bases, bases_mont, scalars, n, algo, threads, out_xy, out_inf);
    if (curve == ZLO_BN254) return bng2_msm_api(bases, bases_mont, scalars, n, algo, threads, out_xy, out_inf);
    return -1;
}
int zlo_g2_mul_gen(int curve, const uint64_t *k, size_t n, uint64_t *out_xy) {
    if (curve == ZLO_BLS12_381) return blsg2_mul_gen_api(bls_G2_GEN, k, n, out_xy);
    if (curve == ZLO_BN254) return bng2_mul_gen_api(bn_G2_GEN, k, n, out_xy);
    return -1;
}

/* ---------------------------------------------------------------- NTT API */
int zlo_ntt(int curve, uint64_t *data, unsigned log_n, int inverse, int coset, int mont) {
    size_t n = (size_t)1 << log_n;
    if (curve == ZLO_BLS12_381) {
        if (log_n > blsntt_TWO_ADICITY) return -1;
        blsr_t *a = (blsr_t *)data;
        if (!mont) for (size_t i = 0; i < n; i++) blsr_from_canon(&a[i], a[i].l);
        blsntt_transform(a, log_n, inverse, coset);
        if (!mont) for (size_t i = 0; i < n; i++) blsr_to_canon(a[i].l, &a[i]);
        return 0;
    }
    if (curve == ZLO_BN254) {
        if (log_n > bnntt_TWO_ADICITY) return -1;
        bnr_t *a = (bnr_t *)data;
        if (!mont) for (size_t i = 0; i < n; i++) bnr_from_canon(&a[i], a[i].l);
        bnntt_transform(a, log_n, inverse, coset);
        if (!mont) for (size_t i = 0; i < n; i++) bnr_to_canon(a[i].l, &a[i]);
        return 0;
    }
    return -1;
}

/* timed variant for bench.py's cpu_baseline (Montgomery limbs in / out): threads <= 1 = the in-order single-threaded transform
 * (the reference's configuration), threads > 1 = the multi-threaded arrangement of the same transform; *seconds = the transform alone */
int zlo_ntt_ex(int curve, uint64_t *data, unsigned log_n, int inverse, int coset, int threads, double *seconds) {
    const double t0 = omp_get_wtime();
    if (curve == ZLO_BLS12_381) {
        if (log_n > blsntt_TWO_ADICITY) return -1;
        if (threads > 1) blsntt_transform_mt((blsr_t *)data, log_n, inverse, coset, threads);
        else blsntt_transform((blsr_t *)data, log_n, inverse, coset);
    } else if (curve == ZLO_BN254) {
        if (log_n > bnntt_TWO_ADICITY) return -1;
        if (threads > 1) bnntt_transform_mt((bnr_t *)data, log_n, inverse, coset, threads);
        else bnntt_transform((bnr_t *)data, log_n, inverse, coset);
    } else {
        return -1;
    }
    if (seconds) *seconds = omp_get_wtime() - t0;
    return 0;
}

/* ---------------------------------------------------------------- Poseidon (pins Fr mul/add vs reference KAT) */
int zlo_poseidon3(const uint64_t *keys, const uint64_t *mds, int rf, int rp, uint64_t *state) {
    /* openzl-tutorials/src/poseidon.rs:165-222 schedule */
    blsr_t s[3], m[9], k;
    for (int i = 0; i < 3; i++) blsr_from_canon(&s[i], state + 4 * i);
    for (int i = 0; i < 9; i++) blsr_from_canon(&m[i], mds + 4 * i);
    int half = rf / 2;
    for (int rnd = 0; rnd < rf + rp; rnd++) {
        for (int i = 0; i < 3; i++) {
            blsr_from_canon(&k, keys + 4 * (3 * rnd + i));
            blsr_add(&s[i], &s[i], &k);
        }
        int full = rnd < half || rnd >= half + rp;
        for (int i = 0; i < (full ? 3 : 1); i++) {
            blsr_t x2, x4;
            blsr_sqr(&x2, &s[i]);
            blsr_sqr(&x4, &x2);
            blsr_mul(&s[i], &x4, &s[i]);
        }
        blsr_t nx[3];
        for (int i = 0; i < 3; i++) {
            blsr_t acc, t;
            blsr_mul(&acc, &m[3 * i], &s[0]);
            blsr_mul(&t, &m[3 * i + 1], &s[1]);
            blsr_add(&acc, &acc, &t);
            blsr_mul(&t, &m[3 * i + 2], &s[2]);
            blsr_add(&nx[i], &acc, &t);
        }
        memcpy(s, nx, sizeof s);
    }
    for (int i = 0; i < 3; i++) blsr_to_canon(state + 4 * i, &s[i]);
    return 0;
}

#include "zl_oracle_groth16.inc"
