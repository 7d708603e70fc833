/* zl_backend.h -- C ABI of libzl_backend.so: the MI355X (gfx950) MSM / NTT backend for OpenZL's arkworks
 * Groth16 plugin path.
 *
 * The reference defines a Rust trait, not an FFI (openzl_crypto::constraint::ProofSystem,
 * /root/reference/openzl-crypto/src/constraint.rs:31-87, implemented for Groth16<E> at
 * /root/reference/plugins/arkworks/src/groth16.rs:405-467).  One level below `Groth16::prove`
 * (groth16.rs:445-457) the work is done by two upstream entry points that the plugin re-exports
 * (`pub use ec;` lib.rs:28-29, `pub use poly;` lib.rs:70-71, `pub use ff::*;` ff.rs:6):
 *     ark_ec::msm::VariableBaseMSM::multi_scalar_mul(bases, scalars) -> Projective      -> zl_msm (group chosen by the bases handle)
 *     ark_poly::EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place(&mut Vec<F>)    -> zl_ntt
 * and the whole prover call                                                               -> zl_groth16_prove
 * These are the symbols a Rust shim (`extern "C"` block) binds.  Plain pointers and sizes only.
 * THIS header is the drop-in boundary (24 functions: what a binding of plugins/arkworks/src/groth16.rs needs); everything the backend offers beyond it --
 * pipelined / partial MSMs, multi-GPU contexts, prover lanes, the C hooks of the host mirror, point codecs, timing -- is in zl_backend_ext.h.
 *
 * Conventions
 *   - return 0 (ZL_OK) on success, a negative ZL_E* code otherwise; nothing throws or aborts across the ABI;
 *   - the caller owns every host buffer; device-resident base tables are owned by the ctx and named by handle
 *     (a proving key is static per circuit: upload once, prove many times);
 *   - a ctx is bound to one GPU and one HIP stream and is used from one thread at a time; ctxs are independent; a zl_mctx bundles one
 *     ctx per device for the sharded entry points;
 *   - field elements are little-endian arrays of u64 limbs, 4 per Fr / BN254 Fq element, 6 per BLS12-381 Fq
 *     element.  ZL_MONT = limbs are arkworks' in-memory Montgomery form (value*2^(64*limbs) mod p), i.e. what
 *     `Fp256/Fp384.0.0` holds; ZL_CANON(0) = canonical integers (what `into_repr()` yields);
 *   - an affine point is x||y (G1) or x.c0||x.c1||y.c0||y.c1 (G2); the all-zero encoding is the point at
 *     infinity ((0,0) is on neither curve), or, with a stride, arkworks' `infinity: bool` byte may be passed;
 *   - results are canonical affine coordinates + an is_infinity byte: unique, hence bit-comparable with the
 *     reference path (`into_affine()` then `into_repr()`).
 */
#ifndef ZL_BACKEND_H
#define ZL_BACKEND_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C++ callers / the library itself: a FIXED underlying type, so that an out-of-range value handed in by a foreign caller is a value the entry point rejects
 * (ZL_EINVAL) and not undefined behaviour at the first load (UBSan -fsanitize=enum on zl_partials_sum(99, ...), round 5).  Same ABI: int. */
#ifdef __cplusplus
#define ZL_ENUM_INT : int
#else
#define ZL_ENUM_INT
#endif
typedef enum ZL_ENUM_INT { ZL_BLS12_381 = 1, ZL_BN254 = 2 } zl_curve_t;
typedef enum ZL_ENUM_INT { ZL_G1 = 1, ZL_G2 = 2 } zl_group_t;
typedef struct zl_ctx zl_ctx;

enum {
    ZL_OK = 0,
    ZL_EINVAL = -1,   /* bad argument (null pointer, unknown curve, n too large, log_n > two-adicity, ...) */
    ZL_ENOMEM = -2,   /* host or device allocation failed */
    ZL_EHIP = -3,     /* a HIP runtime call or kernel failed (zl_ctx_last_hip_error has the code) */
    ZL_ENODEV = -4,   /* no usable gfx950 device */
    ZL_EHANDLE = -5,  /* unknown / freed bases handle, or handle of the wrong curve/group */
    ZL_ENOTCURVE = -6 /* ZL_CHECK was set and a base point is not on the curve */
};

/* flags */
#define ZL_CANON 0u
#define ZL_MONT 1u    /* inputs (and NTT outputs) are Montgomery limbs */
#define ZL_COSET 2u   /* NTT: coset variant (g = Fr multiplicative generator: 7 BLS12-381, 5 BN254) */
#define ZL_INVERSE 4u /* NTT: inverse transform (scaled by n^-1) */
#define ZL_CHECK 8u   /* bases upload: verify y^2 = x^3 + b on the device */
#define ZL_MONT_IN 16u  /* NTT: only the input is Montgomery (ZL_MONT = both sides); the legs of the distributed transform */
#define ZL_MONT_OUT 32u /* NTT: only the output is Montgomery */

/* ---- context -------------------------------------------------------------------------------------------
 * A zl_ctx is SINGLE-CALLER: its streams, scratch slots, event pool and staging buffers belong to the one call in flight (like a HIP stream, it orders
 * work; it is not a lock).  Calls on one ctx from several threads must be serialised by the caller; independent work runs on independent contexts
 * (several per device are fine).  A second pipelined MSM call entering a ctx that is already inside one returns ZL_EINVAL. */
int zl_ctx_create(zl_ctx** out, int device_id);
void zl_ctx_destroy(zl_ctx* ctx);
/* A second PROVER LANE on the parent's device: its own streams, scratch, events and host workers, and read access to the parent's device-resident objects
 * (bases handles with their window tables, R1CS matrices, hence Groth16 keys compiled or decoded on the parent).  N host threads, one lane each, prove side
 * by side over ONE copy of a proving key -- the counterpart of N threads sharing the reference's `&ProvingContext` (groth16.rs:445-457 takes it by shared
 * reference); measured: two lanes raise the proof throughput by 7 % at 958 465 constraints and 45 % at 14 977.  Rules: fork the root
 * ctx only (a fork of a fork is ZL_EINVAL); while a fork lives, the parent refuses zl_bases_free / zl_bases_precompute / zl_r1cs_free (ZL_EINVAL) -- uploads
 * stay allowed; destroy the forks before the parent.  Each lane is single-caller like any ctx. */
int zl_ctx_fork(zl_ctx* parent, zl_ctx** out);
int zl_ctx_last_hip_error(const zl_ctx* ctx);
const char* zl_strerror(int code);

/* ---- device-resident MSM bases (replaces the `bases: &[G::Affine]` argument of multi_scalar_mul) ------- */
/* xy: n points at `stride_bytes` intervals (0 = packed).  inf_offset >= 0: byte offset inside each record of
 * an arkworks-style `infinity: bool`; -1: infinity is the all-zero encoding. */
int zl_bases_upload(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const void* xy, size_t n, size_t stride_bytes,
                    long inf_offset, unsigned flags, uint64_t* handle_out);
/* Optional, MI355X-sized trade of HBM for work: store 2^(c w) P_i for every window w next to the bases (W x the memory:
 * 12 x 2^24 x 128 B = 25.8 GB for 2^24 BLS12-381 G1 points at c = 22; building it takes ~50 plain MSMs' worth of time, so it pays
 * only for a key that is used many times, like a Groth16 proving key) so that all windows share ONE bucket set and c can grow to 22: 12 instead of
 * 16 mixed additions per point.  c = 0 picks c from n.  MSMs on the handle then use the table; results are unchanged. */
int zl_bases_precompute(zl_ctx* ctx, uint64_t handle, int c);
int zl_bases_free(zl_ctx* ctx, uint64_t handle);

/* ---- MSM (replaces VariableBaseMSM::multi_scalar_mul) --------------------------------------------------- */
/* scalars: n x 4 u64 canonical (< r), host memory; uses bases [first, first+n) of the handle.
 * out_xy: canonical affine coordinates (2 or 4 field elements); *out_inf = 1 if the sum is infinity. */
int zl_msm(zl_ctx* ctx, uint64_t bases, size_t first, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* same, scalars already resident in device memory (HBM) */
int zl_msm_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);

/* ---- NTT (replaces Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place) -------------------- */
/* data: 2^log_n Fr elements x 4 u64, in place, natural order in and out; flags: ZL_MONT, ZL_COSET, ZL_INVERSE */
int zl_ntt(zl_ctx* ctx, zl_curve_t curve, uint64_t* data, unsigned log_n, unsigned flags);
int zl_ntt_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned flags);

/* ---- Groth16 prover (replaces ark_groth16::create_random_proof behind Groth16::<E>::prove, groth16.rs:445-457) - */
/* R1CS in CSR form, as ark-relations' ConstraintMatrices hold it: variable order = instance block (index 0 is the
 * constant ONE, then the public inputs) followed by the witness block; coefficients and assignment are canonical
 * Fr integers, 4 u64 each.  [0] = A, [1] = B, [2] = C. */
typedef struct zl_r1cs {
    uint32_t n_constraints, n_instance, n_witness;
    const uint32_t* row_ptr[3]; /* n_constraints + 1 entries */
    const uint32_t* col[3];
    const uint64_t* val[3];
} zl_r1cs;
/* Proving key = ark_groth16::ProvingKey<E> (ProvingContext<E>, groth16.rs:127-140): the five query vectors are
 * device-resident bases handles (zl_bases_upload), the single points canonical affine host buffers. */
typedef struct zl_g16_pk {
    zl_curve_t curve;
    uint64_t a_query, b_g1_query, h_query, l_query; /* ZL_G1 handles: m+1, m+1, N-1, n_witness points */
    uint64_t b_g2_query;                            /* ZL_G2 handle: m+1 points */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1;  /* G1: x||y */
    const uint64_t *beta_g2, *delta_g2;             /* G2: x.c0||x.c1||y.c0||y.c1 */
} zl_g16_pk;
typedef struct zl_g16_proof {
    uint64_t a[12], b[24], c[12]; /* canonical affine A (G1), B (G2), C (G1); BN254 uses the first 8/16/8 words */
    uint8_t a_inf, b_inf, c_inf;
} zl_g16_proof;
/* assignment: (n_instance + n_witness) x 4 u64 canonical (z = 1, public..., witness...); r, s: the two blinding
 * scalars ark samples from the rng (explicit here so that proofs are reproducible, SURVEY.md §8 note N3).
 * Runs witness_map (3 iFFT, 3 coset FFT, 1 coset iFFT on the device) and the 4 G1 + 1 G2 MSMs. */
int zl_groth16_prove(zl_ctx* ctx, const zl_g16_pk* pk, const zl_r1cs* cs, const uint64_t* assignment, const uint64_t* r,
                     const uint64_t* s, zl_g16_proof* out);
/* The matrices are static per circuit: upload them once (like the proving key) and prove many witnesses. */
int zl_r1cs_upload(zl_ctx* ctx, zl_curve_t curve, const zl_r1cs* cs, uint64_t* handle_out);
int zl_r1cs_free(zl_ctx* ctx, uint64_t handle);
/* flags: ZL_MONT = the assignment is in arkworks' in-memory Montgomery form (no host-side into_repr pass needed) */
int zl_groth16_prove_resident(zl_ctx* ctx, const zl_g16_pk* pk, uint64_t r1cs_handle, const uint64_t* assignment, unsigned flags,
                              const uint64_t* r, const uint64_t* s, zl_g16_proof* out);
typedef struct zl_g16_keys zl_g16_keys; /* Groth16<E>::ProvingContext (+ the setup trapdoor, kept for exponent checks) */
void zl_groth16_keys_free(zl_g16_keys* k); /* must be called BEFORE zl_ctx_destroy of the ctx the keys were compiled on (they hold handles of it) */
int zl_groth16_keys_pk(const zl_g16_keys* k, zl_g16_pk* pk);

/* The proof points are taken as given: callers that accept proofs from outside deserialize them with zl_groth16_proof_from_bytes, which
 * checks curve membership and the subgroup (as arkworks' deserialization does); A or B at infinity is rejected here.
 * Groth16::verify (host pairing; public_inputs: n x 4 u64 canonical, without the leading ONE): *ok = 1 accepted, 0 rejected */
int zl_groth16_verify(const zl_g16_keys* k, const uint64_t* public_inputs, size_t n, const zl_g16_proof* proof, int* ok);
size_t zl_groth16_proof_bytes(zl_curve_t curve);
int zl_groth16_proof_to_bytes(zl_curve_t curve, const zl_g16_proof* proof, uint8_t* out);
int zl_groth16_proof_from_bytes(zl_curve_t curve, const uint8_t* in, size_t len, zl_g16_proof* proof);
/* ProvingContext<E> on the wire (its codec::Encode / Decode, /root/reference/plugins/arkworks/src/groth16.rs:142-179): the
 * ark_groth16::ProvingKey<E> written with serialize_unchecked -- vk (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1), beta_g1,
 * delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query; every point uncompressed, every Vec prefixed with its u64 little-endian
 * length.  Same caveat as above: restated from the published ark-serialize / ark-groth16 0.3 layout, not checked against arkworks bytes.
 * to_bytes: *len receives the size; out == NULL with cap == 0 only queries it; cap < size with a buffer is ZL_EINVAL.  The queries are
 * downloaded from the device.
 * from_bytes: uploads the five queries to ctx (flags: 0, or ZL_CHECK = the device verifies the curve equation of every query point),
 * builds the window tables a compiled key of that size gets, and returns keys that prove / verify like compiled ones; they hold no
 * trapdoor (zl_groth16_keys_trapdoor -> ZL_EINVAL) and learn their circuit at the first zl_groth16_prove_circuit (which uploads the
 * matrices and rejects a circuit whose variable counts or evaluation domain do not match the key).  ZL_EINVAL = malformed input. */
int zl_groth16_keys_to_bytes(const zl_g16_keys* k, uint8_t* out, size_t cap, size_t* len);
int zl_groth16_keys_from_bytes(zl_ctx* ctx, zl_curve_t curve, const uint8_t* in, size_t len, unsigned flags, zl_g16_keys** out);

#ifdef __cplusplus
}
#endif
#endif
