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
/* run all work of this ctx on the caller's HIP stream (e.g. torch's current stream); NULL = the ctx's own */
int zl_ctx_set_stream(zl_ctx* ctx, void* hip_stream);
int zl_ctx_sync(zl_ctx* ctx);
/* Pippenger window width c (bits); 0 = choose from n.  Results do not depend on it. */
int zl_ctx_set_msm_window(zl_ctx* ctx, int c);
int zl_ctx_last_hip_error(const zl_ctx* ctx);
const char* zl_strerror(int code);
/* library / device description: writes a NUL-terminated string, returns its length */
int zl_describe(zl_ctx* ctx, char* buf, size_t buflen);

/* ---- device-resident MSM bases (replaces the `bases: &[G::Affine]` argument of multi_scalar_mul) ------- */
/* xy: n points at `stride_bytes` intervals (0 = packed).  inf_offset >= 0: byte offset inside each record of
 * an arkworks-style `infinity: bool`; -1: infinity is the all-zero encoding. */
int zl_bases_upload(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const void* xy, size_t n, size_t stride_bytes,
                    long inf_offset, unsigned flags, uint64_t* handle_out);
/* bases[i] = k[i] * generator, computed on the device (k: n x 4 u64 canonical, host memory).  Input generator
 * for tests and benches: gives MSM inputs with known discrete logs. */
int zl_bases_generate(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const uint64_t* k, size_t n, uint64_t* handle_out);
/* Optional, MI355X-sized trade of HBM for work: store 2^(c w) P_i for every window w next to the bases (W x the memory:
 * 12 x 2^24 x 128 B = 25.8 GB for 2^24 BLS12-381 G1 points at c = 22; building it takes ~50 plain MSMs' worth of time, so it pays
 * only for a key that is used many times, like a Groth16 proving key) so that all windows share ONE bucket set and c can grow to 22: 12 instead of
 * 16 mixed additions per point.  c = 0 picks c from n.  MSMs on the handle then use the table; results are unchanged. */
int zl_bases_precompute(zl_ctx* ctx, uint64_t handle, int c);
/* copy bases back as canonical affine x||y (tests) */
int zl_bases_download(zl_ctx* ctx, uint64_t handle, size_t first, size_t count, uint64_t* out_xy);
int zl_bases_free(zl_ctx* ctx, uint64_t handle);

/* ---- MSM (replaces VariableBaseMSM::multi_scalar_mul) --------------------------------------------------- */
/* scalars: n x 4 u64 canonical (< r), host memory; uses bases [first, first+n) of the handle.
 * out_xy: canonical affine coordinates (2 or 4 field elements); *out_inf = 1 if the sum is infinity. */
int zl_msm(zl_ctx* ctx, uint64_t bases, size_t first, const uint64_t* scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* same, scalars already resident in device memory (HBM) */
int zl_msm_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_xy, uint8_t* out_inf);
/* multi-GPU building block: the un-normalised partial sum of this shard (opaque, ZL_PARTIAL_WORDS u64s), to
 * be all-gathered (RCCL/ncclUint64) and folded with zl_partials_sum on any rank. */
#define ZL_PARTIAL_WORDS 64
int zl_msm_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* d_scalars, size_t n, uint64_t* out_partial);
/* `count` MSMs over the same bases range (e.g. the several scalar vectors of one proof, or a stream of proofs against one proving key):
 * d_scalars[i] points to n x 4 u64 canonical scalars in HBM; out_partials receives count x ZL_PARTIAL_WORDS words.  The calls are
 * pipelined on three streams -- the sort of MSM i+2, the bucket accumulation of MSM i+1 and the merge / reduction tail of MSM i
 * overlap -- so that in steady state an MSM costs little more than its accumulation kernel; results are identical to `count` separate calls. */
int zl_msm_batch_partial_dev(zl_ctx* ctx, uint64_t bases, size_t first, const void* const* d_scalars, size_t n, size_t count, uint64_t* out_partials);
int zl_partials_sum(zl_curve_t curve, zl_group_t group, const uint64_t* partials, size_t count, uint64_t* out_xy, uint8_t* out_inf);
/* wrap a canonical affine point (all-zero = infinity) as a partial, e.g. to fold an extra term into zl_partials_sum */
int zl_partial_from_affine(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint64_t* out_partial);

/* ---- NTT (replaces Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place) -------------------- */
/* data: 2^log_n Fr elements x 4 u64, in place, natural order in and out; flags: ZL_MONT, ZL_COSET, ZL_INVERSE */
int zl_ntt(zl_ctx* ctx, zl_curve_t curve, uint64_t* data, unsigned log_n, unsigned flags);
int zl_ntt_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned flags);
/* Multi-GPU transform (ark-poly has no distributed form -- same function as zl_ntt on the 2^log_n
 * domain, split over G = 2^log_g ranks, M = N/G elements each, B = M/G, with ONE all-to-all between the two local steps):
 *   block-column layout: rank g holds x[j1*M + g*B + c] at local [j1*B + c]  (j1 < G, c < B)
 *   cyclic layout:       rank k holds X[k + G*k2]       at local [k2]        (k2 < M)
 *   forward: zl_ntt_cross_dev (block-column data) -> all-to-all of G chunks of B elements -> zl_ntt_dev(log_n - log_g)
 *            leaves the evaluations in cyclic layout;
 *   inverse: zl_ntt_dev(log_n - log_g, ZL_INVERSE) on cyclic data -> all-to-all -> zl_ntt_cross_dev(ZL_INVERSE)
 *            leaves the coefficients in block-column layout.
 * zl_ntt_cross_dev does the G-point transform across the G local rows of each column, the w_N^(j2 k1) twiddle, and the
 * ZL_COSET scaling of the whole 2^log_n domain; the local M-point leg is always called WITHOUT ZL_COSET.  Elements are
 * Montgomery between the legs (ZL_MONT_IN / ZL_MONT_OUT pick the outer representation).  1 <= log_g <= 4, 2*log_g <= log_n.
 * openzl_amd/sharded.py drives the three steps over torch.distributed (RCCL all_to_all_single). */
int zl_ntt_cross_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned log_g, unsigned rank, unsigned flags);

/* ---- multi-GPU in one process: G devices, one zl_ctx each, RCCL (ncclCommInitAll) between them ------------------
 * Upload / generate / precompute every rank's shard of the bases on ITS ctx (zl_mctx_ctx(m, rank)) with the single-device calls above.
 * device_ids may repeat (several "virtual ranks" on one GPU, for testing on a 1-GPU box): RCCL refuses duplicate devices, so the
 * exchanges then run as device-to-device copies with the same data movement pattern; zl_mctx_uses_rccl tells which.  1 <= n_devices <= 16.
 * Callers that run one process per GPU instead use zl_msm_partial_dev + their own all-gather + zl_partials_sum (openzl_amd/sharded.py). */
typedef struct zl_mctx zl_mctx;
int zl_ctx_create_multi(zl_mctx** out, const int* device_ids, int n_devices);
void zl_mctx_destroy(zl_mctx* m);
int zl_mctx_size(const zl_mctx* m);
zl_ctx* zl_mctx_ctx(zl_mctx* m, int rank);
int zl_mctx_uses_rccl(const zl_mctx* m);
int zl_mctx_last_rccl_error(const zl_mctx* m);
/* Sharded MSM (config 4: 2^26 as 8 x 2^23): rank g holds bases handle bases[g] on its ctx and n[g] canonical scalars in ITS device memory
 * (d_scalars[g]); first may be NULL.  Every device runs the complete local Pippenger concurrently, the G un-normalised partial sums
 * (ZL_PARTIAL_WORDS u64 each) are all-gathered (ncclAllGather as ncclUint64; EC addition is no RCCL reduction op, so gather-then-add
 * IS the reduce) and folded: out_xy = canonical affine sum over all shards. */
int zl_msm_sharded(zl_mctx* m, const uint64_t* bases, const size_t* first, const void* const* d_scalars, const size_t* n, uint64_t* out_xy,
                   uint8_t* out_inf);
/* ONE 2^log_n transform over G = 2^log_g ranks (layouts and legs: zl_ntt_cross_dev below): d_data[g] = rank g's M = 2^(log_n - log_g)
 * elements in its device memory, in place.  forward: block-column coefficients in, cyclic evaluations out; ZL_INVERSE: the reverse.
 * One all-to-all (grouped ncclSend / ncclRecv of G chunks of M/G elements per rank) between the two local legs.  flags: ZL_MONT,
 * ZL_COSET, ZL_INVERSE. */
int zl_ntt_sharded(zl_mctx* m, zl_curve_t curve, void* const* d_data, unsigned log_n, unsigned flags);

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
/* ONE proof over the G devices of an mctx: the five MSMs of create_proof_with_assignment shard by point range like any MSM.  Rank g holds, on
 * zl_mctx_ctx(m, g), bases handles with ITS contiguous slice of every query -- a / b_g1 / b_g2 over the variables [var_first, var_first + var_count) (variable 0
 * is the constant ONE), l over the witnesses [wit_first, ...), h over the domain indices [h_first, ...) of the N - 1 quotient coefficients -- uploaded with the
 * single-device calls; the slices of consecutive ranks must tile the three ranges in rank order (a rank may hold empty slices: count 0, handle ignored).
 * Rank 0 also holds the constraint matrices (r1cs_handle_rank0 from zl_r1cs_upload on zl_mctx_ctx(m, 0)) and runs the witness map; the other ranks receive
 * their slices of z and h by device-to-device copy, every rank runs its five partial MSMs, the host folds the partials over the ranks and assembles the proof.
 * pk: only curve and the five single points are read.  The proof equals zl_groth16_prove_resident's for the same (r, s), byte for byte. */
typedef struct zl_g16_shard {
    uint64_t a_query, b_g1_query, h_query, l_query; /* ZL_G1 handles on this rank's ctx: var_count, var_count, h_count, wit_count points */
    uint64_t b_g2_query;                            /* ZL_G2 handle: var_count points */
    size_t var_first, var_count, wit_first, wit_count, h_first, h_count;
} zl_g16_shard;
int zl_groth16_prove_sharded(zl_mctx* m, const zl_g16_pk* pk, const zl_g16_shard* shards, uint64_t r1cs_handle_rank0, const uint64_t* assignment,
                             unsigned flags, const uint64_t* r, const uint64_t* s, zl_g16_proof* out);
/* the quotient polynomial h of the last successful zl_groth16_prove* call on this ctx (N x 4 u64 canonical), for tests;
 * ZL_EINVAL when there is none (it lives in scratch slot 8 and is invalidated when the next proof starts) */
int zl_groth16_last_h(zl_ctx* ctx, uint64_t* out, size_t n);

/* ---- host mirror of the plugin interface (C hooks over the C++ classes of openzl_amd/csrc/zl_host.h) ------------
 * openzl::R1CS<F> / poseidon gadget / Groth16<E>::{compile, prove} restated in C++ (the reference is Rust; no Rust
 * toolchain here).  These hooks let a C / ctypes caller drive them; a C++ caller uses the classes directly. */
typedef struct zl_circuit zl_circuit;   /* an R1CS<F> compiler in proof mode holding the config-5 circuit */
typedef struct zl_g16_keys zl_g16_keys; /* Groth16<E>::ProvingContext (+ the setup trapdoor, kept for exponent checks) */
/* k chained Poseidon arity-2 hashes over the curve's Fr: h_1 = H(x0,x1), h_{j+1} = H(h_j,x1), public input h_k */
int zl_circuit_poseidon_chain(zl_curve_t curve, uint32_t k, const uint64_t* x0, const uint64_t* x1, zl_circuit** out);
/* The same circuit run by a WITNESS-ONLY compiler (ark-relations' SynthesisMode::Prove { construct_matrices: false }): variables with their values, no linear
 * combinations, no constraint rows -- ~10x faster to synthesise.  zl_groth16_prove_circuit takes it for keys that already hold the circuit's matrices (compiled
 * for it, or bound by an earlier proof with a full circuit); zl_circuit_export then yields an empty CSR and the assignment, zl_circuit_is_satisfied checks only
 * the enforced equalities. */
int zl_circuit_poseidon_chain_witness(zl_curve_t curve, uint32_t k, const uint64_t* x0, const uint64_t* x1, zl_circuit** out);
void zl_circuit_free(zl_circuit* c);
/* CSR view + assignment (pointers stay valid until zl_circuit_free) */
int zl_circuit_export(const zl_circuit* c, zl_r1cs* view, const uint64_t** assignment);
int zl_circuit_is_satisfied(const zl_circuit* c); /* 1 / 0 */
/* native Poseidon permutation, width 3 (tutorial schedule): state = 3 x 4 u64 canonical, in place */
int zl_poseidon_permute(zl_curve_t curve, uint64_t* state);
/* Groth16::compile with rng = SplitMix64(seed): trapdoor setup, proving key generated on the device */
int zl_groth16_compile(zl_ctx* ctx, const zl_circuit* c, uint64_t seed, zl_g16_keys** out);
void zl_groth16_keys_free(zl_g16_keys* k); /* must be called BEFORE zl_ctx_destroy of the ctx the keys were compiled on (they hold handles of it) */
int zl_groth16_keys_pk(const zl_g16_keys* k, zl_g16_pk* pk);
int zl_groth16_keys_trapdoor(const zl_g16_keys* k, uint64_t* out20); /* alpha, beta, gamma, delta, tau (canonical) */
/* Groth16::prove with rng = SplitMix64(seed); r_out / s_out (optional) receive the sampled blinding scalars */
int zl_groth16_prove_circuit(zl_ctx* ctx, const zl_g16_keys* k, const zl_circuit* c, uint64_t seed, zl_g16_proof* proof,
                             uint64_t* r_out, uint64_t* s_out);

/* A STREAM of proofs over one key: proofs[i] = Groth16::prove(keys, *circuits[i], SplitMix64(seeds[i])), i < count, issued from two host threads over two
 * prover lanes (ctx itself and a fork of it that the ctx keeps for later calls) -- the proofs are the ones `count` calls of zl_groth16_prove_circuit return,
 * byte for byte, at the throughput of two lanes (958 465 constraints: 17.7 instead of 18.8 ms per proof; 14 977: 2.35 instead of 3.05; 235: 0.9 instead of
 * 1.5).  ctx must be the root ctx the keys live on and the keys must be bound to their circuit (compiled here, or proven once after decoding); circuits may
 * repeat.  The first failure is returned and the remaining proofs are not started.  The kept lane is the library's own fork and pins nothing once the call
 * has returned: zl_bases_free / zl_bases_precompute / zl_r1cs_free on the ctx release it first when it is idle (round 6; forks the CALLER made with
 * zl_ctx_fork still pin the parent's handles).  zl_ctx_drop_lanes(ctx) releases it explicitly (zl_ctx_destroy does too); it returns ZL_EINVAL while a call is
 * using the lane. */
int zl_groth16_prove_circuits(zl_ctx* ctx, const zl_g16_keys* k, const zl_circuit* const* circuits, const uint64_t* seeds, size_t count, zl_g16_proof* proofs);
int zl_ctx_drop_lanes(zl_ctx* ctx);

/* The proof points are taken as given: callers that accept proofs from outside deserialize them with zl_groth16_proof_from_bytes, which
 * checks curve membership and the subgroup (as arkworks' deserialization does); A or B at infinity is rejected here.
 * Groth16::verify (host pairing; public_inputs: n x 4 u64 canonical, without the leading ONE): *ok = 1 accepted, 0 rejected */
int zl_groth16_verify(const zl_g16_keys* k, const uint64_t* public_inputs, size_t n, const zl_g16_proof* proof, int* ok);
/* e(P, Q) in GT after the final exponentiation: 12 canonical Fq coefficients (BLS12-381: 12 x 6 u64, BN254: 12 x 4 u64) of the
 * polynomial in w, Fq12 = Fq[w]/(w^12 - 2 w^6 + 2) (BLS12-381) or (w^12 - 18 w^6 + 82) (BN254) */
int zl_pairing(zl_curve_t curve, const uint64_t* p_xy, const uint64_t* q_xy, uint64_t* out12);

/* ---- wire formats: arkworks 0.3 CanonicalSerialize, compressed (replaces proof_as_bytes / HasSerialization,
 * /root/reference/plugins/arkworks/src/groth16.rs:68-107).  x little-endian, flags in the top two bits of the last byte
 * (bit 7: y is the larger root, bit 6: infinity); Fq2: c0 then c1 (flags on c1).  Proof = A || B || C: 192 bytes (BLS12-381) / 128 (BN254).
 * Host only.  NOT verified against arkworks-produced bytes (the reference holds no vector): pinned to an independent restatement
 * (oracle/pyoracle.py) and round trips.  from_bytes: ZL_EINVAL = malformed, ZL_ENOTCURVE = no such point / outside the subgroup. */
size_t zl_point_bytes(zl_curve_t curve, zl_group_t group);
int zl_point_to_bytes(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint8_t inf, uint8_t* out);
int zl_point_from_bytes(zl_curve_t curve, zl_group_t group, const uint8_t* in, uint64_t* xy, uint8_t* inf);
size_t zl_groth16_proof_bytes(zl_curve_t curve);
int zl_groth16_proof_to_bytes(zl_curve_t curve, const zl_g16_proof* proof, uint8_t* out);
int zl_groth16_proof_from_bytes(zl_curve_t curve, const uint8_t* in, size_t len, zl_g16_proof* proof);
/* Uncompressed form (ark's serialize_uncompressed / serialize_unchecked): x then y, flags on the last byte of y; a finite point
 * carries no flag bits, infinity is written as (0, 1) with bit 6 set.  check = 0 is deserialize_unchecked (coordinates must be canonical
 * integers, nothing else is verified); check != 0 also requires a point of the prime-order subgroup (ZL_ENOTCURVE otherwise). */
size_t zl_point_bytes_uncompressed(zl_curve_t curve, zl_group_t group);
int zl_point_to_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint8_t inf, uint8_t* out);
int zl_point_from_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint8_t* in, int check, uint64_t* xy, uint8_t* inf);
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
/* The same bytes validated on the HOST only (no ctx, no device memory): framing, Vec lengths against the input size, canonical coordinates, flag bits, the shape
 * relations between the five queries; ZL_CHECK also verifies the verifying-key points (curve and subgroup).  ZL_OK = zl_groth16_keys_from_bytes would accept the
 * framing (its device-side curve checks of the queries under ZL_CHECK are not repeated here).  Same error codes. */
int zl_groth16_keys_parse(zl_curve_t curve, const uint8_t* in, size_t len, unsigned flags);
/* ark_groth16::VerifyingKey<E>::serialize (compressed points): alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 (u64 length first).
 * The reference's VerifyingContext (/root/reference/plugins/arkworks/src/groth16.rs:181-396) frames a PreparedVerifyingKey as
 *     vk | alpha_g1_beta_g2 (one Fqk) | gamma_g2_neg_pc | delta_g2_neg_pc
 * where the last two are written through `<E::G2Prepared as HasSerialization>::Serialize` (groth16.rs:208-211).  HasSerialization /
 * HasDeserialization are hook traits (plugins/arkworks/src/serialize.rs:21-30) that NO file of the reference implements for any curve
 * (grep: no `impl ... HasSerialization`): the bytes of a prepared G2 point (ark-ec's Miller-loop line coefficients) are left to a
 * downstream crate, so the reference defines no layout for two of the four fields and VerifyingContext: Encode / Decode cannot even be
 * instantiated from the reference alone.  alpha_g1_beta_g2 is the pairing value ark's final exponentiation produces; this backend's
 * pairing (csrc/zl_pairing.h) takes the exact (q^12 - 1)/r power in a tower-free basis, and whether ark-ec 0.3's hard part returns that
 * power or a fixed multiple of it cannot be checked without the crate.  What IS defined and restated is the first field, vk, below; a
 * verifier built from it recomputes the other three (zl_groth16_verify does). */
int zl_groth16_vk_to_bytes(const zl_g16_keys* k, uint8_t* out, size_t cap, size_t* len);

/* ---- per-call device timing (HIP events on the ctx's stream) -------------------------------------------- */
typedef struct zl_timing {
    float total_ms;      /* first kernel start -> last kernel end of the last zl_msm* / zl_ntt* call */
    float dominant_ms;   /* the dominant kernel: bucket accumulation (MSM) / all butterfly passes (NTT) */
    uint32_t launches;   /* launches of the dominant kernel in that call */
    uint32_t window_bits;
    uint64_t entries;    /* MSM: (point, window) pairs accumulated */
} zl_timing;
int zl_ctx_enable_timing(zl_ctx* ctx, int on);
int zl_last_timing(zl_ctx* ctx, zl_timing* out);

#ifdef __cplusplus
}
#endif
#endif
