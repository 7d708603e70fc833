/* zl_backend_ext.h -- the rest of libzl_backend.so's C ABI: everything beyond the boundary the reference's plugin binds (include/zl_backend.h).
 *
 * Split off in round 6: zl_backend.h holds the drop-in boundary -- the symbols a binding of
 * /root/reference/plugins/arkworks/src/groth16.rs needs: context, bases, zl_msm, zl_ntt, zl_groth16_prove*, the key / proof codecs, verify.  This header
 * holds what the backend offers on top: device-resident and pipelined MSM entry points, partial sums for sharding, the multi-GPU contexts
 * (zl_mctx, zl_msm_sharded, zl_ntt_sharded, zl_groth16_prove_sharded), prover lanes, the C hooks of the C++ host mirror (circuits, compile, prove_circuit),
 * the host pairing, point codecs and timing.  Same conventions, same library. */
#ifndef ZL_BACKEND_EXT_H
#define ZL_BACKEND_EXT_H
#include "zl_backend.h"
#ifdef __cplusplus
extern "C" {
#endif
/* run all work of this ctx on the caller's HIP stream (e.g. torch's current stream); NULL = the ctx's own */
int zl_ctx_set_stream(zl_ctx* ctx, void* hip_stream);
int zl_ctx_sync(zl_ctx* ctx);
/* Pippenger window width c (bits); 0 = choose from n.  Results do not depend on it. */
int zl_ctx_set_msm_window(zl_ctx* ctx, int c);
/* library / device description: writes a NUL-terminated string, returns its length */
int zl_describe(zl_ctx* ctx, char* buf, size_t buflen);
/* bases[i] = k[i] * generator, computed on the device (k: n x 4 u64 canonical, host memory).  Input generator
 * for tests and benches: gives MSM inputs with known discrete logs. */
int zl_bases_generate(zl_ctx* ctx, zl_curve_t curve, zl_group_t group, const uint64_t* k, size_t n, uint64_t* handle_out);
/* copy bases back as canonical affine x||y (tests) */
int zl_bases_download(zl_ctx* ctx, uint64_t handle, size_t first, size_t count, uint64_t* out_xy);
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
/* `count` transforms of the same size and flags in one launch per pass: vector v starts at d_data + v * stride_elems elements (stride_elems >= 2^log_n).  What the
 * Groth16 witness map does with its a, b, c vectors (groth16.rs:445-457 -> ark-groth16's witness map: three iffts, three coset ffts): below ~2^16 elements a transform
 * is a chain of latency-bound passes, and three of them cost what one does.  Results are those of `count` zl_ntt_dev calls. */
int zl_ntt_batch_dev(zl_ctx* ctx, zl_curve_t curve, void* d_data, unsigned log_n, unsigned flags, unsigned count, size_t stride_elems);

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
/* Uncompressed form (ark's serialize_uncompressed / serialize_unchecked): x then y, flags on the last byte of y; a finite point
 * carries no flag bits, infinity is written as (0, 1) with bit 6 set.  check = 0 is deserialize_unchecked (coordinates must be canonical
 * integers, nothing else is verified); check != 0 also requires a point of the prime-order subgroup (ZL_ENOTCURVE otherwise). */
size_t zl_point_bytes_uncompressed(zl_curve_t curve, zl_group_t group);
int zl_point_to_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint64_t* xy, uint8_t inf, uint8_t* out);
int zl_point_from_bytes_uncompressed(zl_curve_t curve, zl_group_t group, const uint8_t* in, int check, uint64_t* xy, uint8_t* inf);
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
