/* oracle/zl_oracle.h -- C API of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * CPU restatement of the arkworks 0.3.x algorithms that OpenZL's plugins/arkworks Groth16 backend
 * delegates to (reference call sites plugins/arkworks/src/groth16.rs:438,454).  It is the checker
 * for the HIP path and the timed "port" cpu_baseline.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library never links or calls it.
 *
 * PARITY STATUS: unpinned at the MSM / NTT / Groth16 boundary (the reference holds no vector for
 * them); pinned for BLS12-381 Fr arithmetic by the reference's Poseidon fixtures
 * (tests/golden/ref_poseidon_fixtures.json) and cross-checked against the independent big-int
 * model oracle/pyoracle.py.  Data layouts are those of include/zl_backend.h.
 */
#ifndef ZL_ORACLE_H
#define ZL_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZLO_BLS12_381 1
#define ZLO_BN254 2
/* field ids */
#define ZLO_F_BLS_FQ 1
#define ZLO_F_BLS_FR 2
#define ZLO_F_BN_FQ 3
#define ZLO_F_BN_FR 4

int zlo_threads_available(void);

/* r = a*b (op 0), a+b (1), a-b (2), a^-1 (3; b ignored), over field `fid`; canonical LE limbs in/out
 * (4 or 6 u64).  Goes through the Montgomery path, so it pins the same code MSM/NTT use. */
int zlo_field_op(int fid, int op, const uint64_t *a, const uint64_t *b, uint64_t *r);
/* canonical <-> Montgomery limbs (n elements) */
int zlo_field_to_mont(int fid, const uint64_t *in, uint64_t *out, size_t n);
int zlo_field_from_mont(int fid, const uint64_t *in, uint64_t *out, size_t n);

/* G1 MSM.  bases: n x (x||y) coordinate limbs (4 u64 BN254 / 6 u64 BLS12-381 each), all-zero pair = infinity;
 * bases_mont: 1 if coordinates are Montgomery limbs, 0 if canonical.  scalars: n x 4 u64 canonical.
 * algo: 0 = ark Pippenger, 1 = definition (sum of double-and-add).  threads: <=1 single thread.
 * out_xy: canonical affine x||y; *out_inf = 1 when the sum is infinity (out_xy zeroed). */
int zlo_msm_g1(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo,
               int threads, uint64_t *out_xy, uint8_t *out_inf);
/* Same with timing for bench.py's cpu_baseline: the bases are loaded in parallel (untimed), *seconds = the MSM alone.
 * algo 0: ark Pippenger, window-parallel over `threads` (c_override > 0 forces the window width, e.g. the width ark's rule gives the
 * full-size workload when a bounded sample is timed); algo 2: point-chunked over `threads` (every thread runs the single-threaded
 * ark algorithm on its chunk; not an arkworks configuration; uses every core, not necessarily the fastest arrangement: bench.py reports it beside the window-parallel run). */
int zlo_msm_g1_ex(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo, int c_override,
                  int threads, uint64_t *out_xy, uint8_t *out_inf, double *seconds);
/* G2 MSM: bases n x (x.c0||x.c1||y.c0||y.c1). */
int zlo_msm_g2(int curve, const uint64_t *bases, int bases_mont, const uint64_t *scalars, size_t n, int algo,
               int threads, uint64_t *out_xy, uint8_t *out_inf);
/* out[i] = k[i]*G (canonical affine x||y), k: n x 4 u64 canonical.  Used to make test bases. */
int zlo_g1_mul_gen(int curve, const uint64_t *k, size_t n, uint64_t *out_xy);
int zlo_g2_mul_gen(int curve, const uint64_t *k, size_t n, uint64_t *out_xy);
/* out = k*P for one affine canonical point */
int zlo_g1_mul(int curve, const uint64_t *p_xy, const uint64_t *k, uint64_t *out_xy, uint8_t *out_inf);

/* NTT over the scalar field of `curve`, in place, natural order in/out; data n=2^log_n x 4 u64;
 * mont: 1 = Montgomery limbs in/out, 0 = canonical in/out. */
int zlo_ntt(int curve, uint64_t *data, unsigned log_n, int inverse, int coset, int mont);

/* timed variant (Montgomery limbs): threads <= 1 single-threaded in-order transform, > 1 the same transform split over threads */
int zlo_ntt_ex(int curve, uint64_t *data, unsigned log_n, int inverse, int coset, int threads, double *seconds);

/* Poseidon permutation width 3 (tutorial schedule) over BLS12-381 Fr from caller-supplied constants:
 * keys (3*(rf+rp) canonical), mds (9 canonical, row-major), state 3 canonical in/out. */
int zlo_poseidon3(const uint64_t *keys, const uint64_t *mds, int rf, int rp, uint64_t *state);

/* ---- Groth16 prove (ark-groth16 0.3.0 create_proof_with_assignment + R1CStoQAP::witness_map) ----------------
 * R1CS in CSR form (A, B, C), variable order = instance block (index 0 is the constant ONE) then witnesses;
 * coefficients and assignment canonical 4 x u64.  Proving key as host arrays of canonical affine points. */
typedef struct zlo_r1cs {
    uint32_t n_constraints, n_instance, n_witness;
    const uint32_t *row_ptr[3];
    const uint32_t *col[3];
    const uint64_t *val[3];
} zlo_r1cs;
typedef struct zlo_g16_pk {
    const uint64_t *a_query, *b_g1_query, *h_query, *l_query; /* G1: x||y */
    const uint64_t *b_g2_query;                               /* G2: x.c0||x.c1||y.c0||y.c1 */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1, *beta_g2, *delta_g2;
} zlo_g16_pk;
typedef struct zlo_g16_proof {
    uint64_t a[12], b[24], c[12]; /* canonical affine, sized for BLS12-381 (BN254 uses the prefix) */
    uint8_t a_inf, b_inf, c_inf;
    uint64_t *h_out; /* optional: N x 4 u64 quotient coefficients (witness_map output), may be NULL */
} zlo_g16_proof;
int zlo_groth16_prove(int curve, const zlo_r1cs *cs, const uint64_t *assignment, const zlo_g16_pk *pk, const uint64_t *r,
                      const uint64_t *s, int threads, zlo_g16_proof *out);

#ifdef __cplusplus
}
#endif
#endif
