// zl_ntt.hip -- radix-2 number-theoretic transform over the scalar field on gfx950.
//
// Replaces ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place (ark-poly 0.3.0; reached from
// /root/reference/plugins/arkworks/src/groth16.rs:454 through R1CStoQAP::witness_map, and surfaced by `pub use poly;`
// /root/reference/plugins/arkworks/src/lib.rs:70-71).  Same contract: natural order in and out,
// group_gen = TWO_ADIC_ROOT^(2^(TWO_ADICITY - log n)), inverse scaled by n^-1, coset variants pre-multiply by g^i /
// post-multiply by g^-i.  arkworks runs log n in-place butterfly sweeps over a tabulated root vector; here the
// transform is factored N = N1*N2*..*NP (each Np <= 2^10, usually 2^8) and every factor is one kernel pass:
//   pass p: a workgroup owns a tile of 2^sp rows x C columns (C*32 B contiguous in HBM), stages it in LDS (limb-major
//           SoA, swizzled), applies the inter-factor twiddle w_N^(j_p * K) on load, runs the 2^sp-point DIF
//           butterflies entirely in LDS with the w_(2^sp) table also in LDS, and stores rows bit-reversed.
//   pass 1 reads the caller's buffer and writes a scratch buffer, the last pass reads scratch and writes the caller's
//   buffer transposed into natural order, so the transform is in place for the caller with no extra copy.
// HBM traffic: P reads + P writes of the vector (P = 3 at 2^24); everything else stays in LDS / registers.
#include <string.h>
#include <algorithm>
#include <vector>
#include "zl_ctx.h"
#include "zl_field28r.h"

// first / last two-stage round fused with the global loads / stores (measured, 2^24: no gain over staging through LDS: the fused kernel needs
// 145-158 VGPRs, one workgroup per CU, 3.0 ms; capped at 128 VGPRs it spills and ties with the unfused form at 2.6 ms)
#ifndef ZL_NTT_FUSE_LOAD
#define ZL_NTT_FUSE_LOAD 0
#endif
#ifndef ZL_NTT_FUSE_STORE
#define ZL_NTT_FUSE_STORE 0
#endif
#define NTT_THREADS 512
#define NTT_TILE 2048   // elements per workgroup tile (64 KiB of LDS)
#define NTT_MAX_S 10

struct NttArgs {
    uint32_t n_log, s, logC, P, p;
    uint32_t sizes[4];
    uint32_t S_prev;        // s_1 + .. + s_(p-1)
    uint32_t L;             // two-level table split: e = hi << L | lo
    uint32_t pre_coset, post_scale, post_coset, to_mont, from_mont;
    const void *t_lo, *t_hi;  // w^lo, w^(hi << L)
    const void* w_small;      // w_(2^s)^i, i < 2^(s-1)
    const void* w_unpacked;   // lazy passes: the same roots as limbs (4 L bytes each), read from global memory instead of LDS
    const void *g_lo, *g_hi;  // coset powers (hi table carries n^-1 for the inverse)
    const void* row_tw;       // middle pass (lazy form): w_N^((r K0(hi)) << shift) for every (hi, r), hi = the tile's high index (or null: combined per tile)
    const void* last_tw;      // last pass of a multi-pass transform: the complete inter-factor twiddle of every element, in load order (or null)
    uint32_t ninv[8];
    uint64_t in_batch, out_batch;  // bytes between the vectors of a batch (blockIdx.y = vector): equal-size transforms of one launch (Groth16's witness map: three at a time)
};

template <class F>
__device__ __forceinline__ F lds_load(const uint32_t* sh, uint32_t pos) {
    F r;
#pragma unroll
    for (int k = 0; k < F::N; k++) r.l[k] = sh[k * NTT_TILE + pos];
    return r;
}
template <class F>
__device__ __forceinline__ void lds_store(uint32_t* sh, uint32_t pos, const F& v) {
#pragma unroll
    for (int k = 0; k < F::N; k++) sh[k * NTT_TILE + pos] = v.l[k];
}
#ifndef ZL_NTT_ROW_SWIZZLE
#define ZL_NTT_ROW_SWIZZLE 0  // measured: conflicts 57 % -> 0, time +1 % (the passes are bound by VALU issue, the LDS cycles were hidden): off; -DZL_NTT_ROW_SWIZZLE=1 rebuilds it
#endif
__device__ __forceinline__ uint32_t tile_pos(uint32_t row, uint32_t col, uint32_t logC) {
    // swizzle columns so that consecutive rows of one column fall into different LDS banks
    uint32_t C = 1u << logC;
    uint32_t rows_per_wrap_log = logC >= 5 ? 0 : 5 - logC;
    uint32_t cs = (col + (row >> rows_per_wrap_log)) & (C - 1);
#if ZL_NTT_ROW_SWIZZLE
    // Round 6 (profiles/r06_ntt_lds_counters.txt: SQ_LDS_BANK_CONFLICT = 57 % of SQ_LDS_IDX_ACTIVE in a 2^8-point pass): with four-column rows (the 2^8 x 4 tiles
    // of every pass of a 2^24 transform) a row owns the four banks (row mod 8) * 4 .. + 3 of the 32 that ds_read2_b32 / ds_write see, and the column swizzle
    // above only permutes inside that group.  The rows a half-wave (32 lanes = 8 rows x 4 columns, the conflict domain) touches at once differ in row bits
    // {0,1,2} (load phase, butterfly distances 64 and 16), {0,1,4} (distance 4), {2,3,4} (distance 1) and {5,6,7} (store phase: bit-reversed rows) -- the
    // last three put 2, 4 and 8 rows on one bank group.  XOR-ing the low three row bits with b0 = r4 ^ r5, b1 = r3 ^ r6, b2 = r4 ^ r7 makes the bank group a
    // GF(2)-linear function of the row whose restriction to each of those four bit sets is invertible: eight distinct groups in every phase; and
    // row -> row ^ f(row >> 3) is a bijection, so the tile stays a permutation of itself.
    if (logC == 2) row ^= (((row >> 4) ^ (row >> 5)) & 1u) | ((((row >> 3) ^ (row >> 6)) & 1u) << 1) | ((((row >> 4) ^ (row >> 7)) & 1u) << 2);
#endif
    return (row << logC) | cs;
}
template <class F>
__device__ __forceinline__ F twiddle2(const F* lo, const F* hi, uint32_t L, uint64_t e) {
    const F a = lo[e & ((1ull << L) - 1)];
    const F b = hi[e >> L];
    return zl::mul(a, b);
}

template <class FrP, bool LAST>
__global__ void __launch_bounds__(NTT_THREADS, 4) k_ntt_pass(const Fp<FrP>* __restrict__ in, Fp<FrP>* __restrict__ out, NttArgs a) {
    using F = Fp<FrP>;
    in = reinterpret_cast<const Fp<FrP>*>(reinterpret_cast<const unsigned char*>(in) + (size_t)blockIdx.y * a.in_batch);
    out = reinterpret_cast<Fp<FrP>*>(reinterpret_cast<unsigned char*>(out) + (size_t)blockIdx.y * a.out_batch);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sh = reinterpret_cast<uint32_t*>(smem);                       // [8][NTT_TILE]
    const uint32_t tid = threadIdx.x;
    const uint32_t s = a.s, logC = a.logC, R = 1u << s, C = 1u << logC;
    F* sh_w = reinterpret_cast<F*>(smem + (size_t)F::N * 4 * NTT_TILE);      // [2^(s-1)] butterfly roots
    F* sh_row = sh_w + (R >> 1);                                             // [2^s] per-row twiddles (non-last passes)
    const uint32_t n_log = a.n_log;
    const F* t_lo = reinterpret_cast<const F*>(a.t_lo);
    const F* t_hi = reinterpret_cast<const F*>(a.t_hi);

    // ---- tile addressing -------------------------------------------------------------------------------
    uint64_t in_base, out_base, in_row, in_col, out_row;
    uint64_t K0 = 0;  // digit-reversed index of the already transformed factors (column 0)
    {
        const uint64_t tile = blockIdx.x;
        if (!LAST) {
            const uint32_t stride_log = n_log - a.S_prev - s;  // elements between consecutive rows
            const uint64_t lo_blocks = (1ull << stride_log) >> logC;
            const uint64_t hi = tile / lo_blocks, lo0 = (tile % lo_blocks) << logC;
            in_base = (hi << (s + stride_log)) + lo0;
            out_base = in_base;
            in_row = 1ull << stride_log;
            in_col = 1;
            out_row = in_row;
            uint64_t rem = hi;
            uint32_t Sq = a.S_prev;
            for (int q = (int)a.p - 2; q >= 0; q--) {  // digits k_(p-1) .. k_1, least significant first
                Sq -= a.sizes[q];
                K0 += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
                rem >>= a.sizes[q];
            }
        } else if (a.P == 1) {
            in_base = out_base = 0;
            in_row = out_row = 1;
            in_col = 1;
        } else {
            const uint32_t s1 = a.sizes[0];
            const uint32_t rest_log = a.S_prev - s1;  // bits of (k_2 .. k_(P-1))
            const uint64_t k1_blocks = (1ull << s1) >> logC;
            const uint64_t k1_0 = (tile % k1_blocks) << logC, rest = tile / k1_blocks;
            in_base = ((k1_0 << rest_log) + rest) << s;
            in_col = 1ull << (rest_log + s);
            in_row = 1;
            uint64_t rem = rest, Krest = 0;
            uint32_t Sq = a.S_prev;
            for (int q = (int)a.P - 2; q >= 1; q--) {
                Sq -= a.sizes[q];
                Krest += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
                rem >>= a.sizes[q];
            }
            K0 = k1_0 + Krest;
            out_base = K0;
            out_row = 1ull << a.S_prev;
        }
    }
    // ---- stage the small tables ----------------------------------------------------------------------------
    {
        const F* w_small = reinterpret_cast<const F*>(a.w_small);
        for (uint32_t i = tid; i < (R >> 1); i += NTT_THREADS) sh_w[i] = w_small[i];
        if (!LAST && a.p > 1) {
            const uint32_t shift = n_log - a.S_prev - s;  // N / M_p
            for (uint32_t r = tid; r < R; r += NTT_THREADS) sh_row[r] = twiddle2(t_lo, t_hi, a.L, ((uint64_t)r * K0) << shift);
        }
    }
    __syncthreads();
    // ---- element access: load = global read + Montgomery entry / coset scaling / inter-factor twiddle; store = output row k of the tile
    //      (+ coset / n^-1 scaling and Montgomery exit on the last pass) ----------------------------------------------------------
    auto load_elem = [&](uint32_t r, uint32_t col) -> F {
        const uint64_t m = in_base + (uint64_t)r * in_row + (uint64_t)col * in_col;
        F x = in[m];
        if (a.p == 1) {
            if (a.to_mont) x = zl::to_mont(x);
            if (a.pre_coset) x = zl::mul(x, twiddle2(reinterpret_cast<const F*>(a.g_lo), reinterpret_cast<const F*>(a.g_hi), a.L, m));
        } else if (!LAST) {
            x = zl::mul(x, sh_row[r]);
        } else if (a.last_tw) {
            x = zl::mul(x, reinterpret_cast<const F*>(a.last_tw)[m]);  // one multiplication: the factor was combined once per key (k_ntt_last_table)
        } else {
            x = zl::mul(x, twiddle2(t_lo, t_hi, a.L, (uint64_t)r * (K0 + col)));
        }
        return x;
    };
    auto store_elem = [&](uint32_t k, uint32_t col, F x) {
        const uint64_t m = out_base + (uint64_t)k * out_row + col;
        if (LAST) {
            if (a.post_coset) {
                x = zl::mul(x, twiddle2(reinterpret_cast<const F*>(a.g_lo), reinterpret_cast<const F*>(a.g_hi), a.L, m));
            } else if (a.post_scale) {
                F ni;
#pragma unroll
                for (int w = 0; w < F::N; w++) ni.l[w] = a.ninv[w];
                x = zl::mul(x, ni);
            }
            if (a.from_mont) x = zl::from_mont(x);
        }
        out[m] = x;
    };
    // two DIF stages on four rows of one column held in registers: stage (distance h) on the pairs (x0, x2), (x1, x3), then stage
    // (distance h2 = h / 2) on (., .) -- rows i0, i0 + h2, i0 + h, i0 + h + h2 with i0 = (blk << (hl + 2)) | k2
    auto quad = [&](F& x0, F& x1, F& x2, F& x3, uint32_t k2, uint32_t hl) {
        const uint32_t h2 = 1u << hl, sh1 = s - 2 - hl;  // stage hl + 1: twiddle index k << (s - 2 - hl); stage hl: k << (s - 1 - hl)
        const F a0 = zl::add(x0, x2), a1 = zl::add(x1, x3);
        F a2 = zl::sub(x0, x2);
        if (k2 != 0) a2 = zl::mul(a2, sh_w[k2 << sh1]);
        const F a3 = zl::mul(zl::sub(x1, x3), sh_w[(k2 + h2) << sh1]);
        x0 = zl::add(a0, a1);
        x2 = zl::add(a2, a3);
        x1 = zl::sub(a0, a1);
        x3 = zl::sub(a2, a3);
        if (k2 != 0) {
            const F w = sh_w[k2 << (sh1 + 1)];
            x1 = zl::mul(x1, w);
            x3 = zl::mul(x3, w);
        }
    };
    // ---- 2^s-point DIF butterflies (all C columns), two stages per round; the first round takes its rows straight from global memory, the
    //      last one (even s >= 4) writes straight to global memory: three LDS round trips and three barriers per pass at s = 8 instead of
    //      ten / nine with one stage per round trip.  (Radix-4 by register blocking only: w_4 is an ordinary field multiplication.)
    const bool fuse_store = ZL_NTT_FUSE_STORE && (s & 1u) == 0 && s >= 4;
    uint32_t hl = s;
    if (ZL_NTT_FUSE_LOAD && s >= 2) {
        hl -= 2;
        const uint32_t h2 = 1u << hl;  // R / 4
        for (uint32_t q = tid; q < (R >> 2) * C; q += NTT_THREADS) {
            uint32_t k2, col;
            if (LAST && a.P > 1) { k2 = q & (h2 - 1); col = q >> hl; } else { col = q & (C - 1); k2 = q >> logC; }  // consecutive lanes: contiguous memory
            F x0 = load_elem(k2, col), x1 = load_elem(k2 + h2, col), x2 = load_elem(k2 + 2 * h2, col), x3 = load_elem(k2 + 3 * h2, col);
            quad(x0, x1, x2, x3, k2, hl);
            lds_store(sh, tile_pos(k2, col, logC), x0);
            lds_store(sh, tile_pos(k2 + h2, col, logC), x1);
            lds_store(sh, tile_pos(k2 + 2 * h2, col, logC), x2);
            lds_store(sh, tile_pos(k2 + 3 * h2, col, logC), x3);
        }
    } else {
        for (uint32_t idx = tid; idx < R * C; idx += NTT_THREADS) {
            const uint32_t col = idx & (C - 1), r = idx >> logC;
            lds_store(sh, tile_pos(r, col, logC), load_elem(r, col));
        }
    }
    __syncthreads();
    while (hl >= (fuse_store ? 4u : 2u)) {
        hl -= 2;
        const uint32_t h2 = 1u << hl, h = h2 << 1;
        for (uint32_t q = tid; q < (R >> 2) * C; q += NTT_THREADS) {
            const uint32_t col = q & (C - 1), rr = q >> logC;
            const uint32_t k2 = rr & (h2 - 1), blk = rr >> hl;
            const uint32_t i0 = (blk << (hl + 2)) | k2;
            const uint32_t p0 = tile_pos(i0, col, logC), p1 = tile_pos(i0 + h2, col, logC), p2 = tile_pos(i0 + h, col, logC),
                           p3 = tile_pos(i0 + h + h2, col, logC);
            F x0 = lds_load<F>(sh, p0), x1 = lds_load<F>(sh, p1), x2 = lds_load<F>(sh, p2), x3 = lds_load<F>(sh, p3);
            quad(x0, x1, x2, x3, k2, hl);
            lds_store(sh, p0, x0);
            lds_store(sh, p1, x1);
            lds_store(sh, p2, x2);
            lds_store(sh, p3, x3);
        }
        __syncthreads();
    }
    if (fuse_store) {
        // last round (distances 2 and 1) from LDS, results straight to global memory: row i0 + j of the tile is output index bitrev_s(i0 + j)
        for (uint32_t q = tid; q < (R >> 2) * C; q += NTT_THREADS) {
            const uint32_t col = q & (C - 1), rr = q >> logC;
            const uint32_t i0 = rr << 2;
            F x0 = lds_load<F>(sh, tile_pos(i0, col, logC)), x1 = lds_load<F>(sh, tile_pos(i0 + 1, col, logC)),
              x2 = lds_load<F>(sh, tile_pos(i0 + 2, col, logC)), x3 = lds_load<F>(sh, tile_pos(i0 + 3, col, logC));
            quad(x0, x1, x2, x3, 0u, 0u);
            store_elem(__brev(i0) >> (32 - s), col, x0);
            store_elem(__brev(i0 + 1) >> (32 - s), col, x1);
            store_elem(__brev(i0 + 2) >> (32 - s), col, x2);
            store_elem(__brev(i0 + 3) >> (32 - s), col, x3);
        }
        return;
    }
    if (hl == 1) {  // odd s: the last stage (distance 1) on its own
        for (uint32_t q = tid; q < (R >> 1) * C; q += NTT_THREADS) {
            const uint32_t col = q & (C - 1), rr = q >> logC;
            const uint32_t i = rr << 1;
            const uint32_t pu = tile_pos(i, col, logC), pv = tile_pos(i + 1, col, logC);
            const F u = lds_load<F>(sh, pu), v = lds_load<F>(sh, pv);
            lds_store(sh, pu, zl::add(u, v));
            lds_store(sh, pv, zl::sub(u, v));
        }
        __syncthreads();
    }
    // ---- store: output row k sits at LDS row bitrev(k) ------------------------------------------------------
    for (uint32_t idx = tid; idx < R * C; idx += NTT_THREADS) {
        const uint32_t col = idx & (C - 1), k = idx >> logC;
        const uint32_t row = s ? (__brev(k) >> (32 - s)) : 0;
        store_elem(k, col, lds_load<F>(sh, tile_pos(row, col, logC)));
    }
}

// ------------------------------------------------------------------------------------------------ the passes on lazily reduced 28-bit limbs
// Round 4: the same pass with the tile held as 10 x 28-bit lazily reduced elements (zl_field28r.h).  A butterfly addition is ten
// v_add_u32 and a carry pass, a subtraction adds a biased multiple 2^j r of the modulus first (no comparison anywhere between two multiplications), a
// product is a carry-free chain of 210 v_mad_u64_u32 under the Montgomery radix R' = 2^280 -- of the MULTIPLIER only: the data keeps the caller's form
// (canonical or R = 2^256 Montgomery; mul(x, w R') = x w), so a canonical -> canonical transform needs no conversion multiplications at all, and the
// tables hold w R' mod r (built in the 32-bit field, converted once: k_ntt_to_lazy).  Global memory keeps its 32-byte elements: unpack on load; on store a
// weak reduction from the top limb (< 2r < 2^256) between passes and the canonical form at the end.
// Bounds: two-stage round r of a pass enters with every element below B_r r; its subtractions use K1 = 2^j r >= (B_r + 2) r (first stage; also the
// second stage's a2 - a3, whose subtrahend is a product < 2r) and K2 >= (2 B_r + 2) r (second stage's a0 - a1); B_(r+1) = max(4 B_r, B_r + K1 + 4, 2 B_r + K2)
// <= 6 B_r + 7: five rounds from B = 6 stay below 2^16 (the multiplier takes 2^25, the top limb 2^29).  The host simulates this and passes the biases.
#ifndef NTT28_TILE_LOG
#define NTT28_TILE_LOG 10  // elements per tile of the lazy passes (A/B: -DNTT28_TILE_LOG=11 -DNTT28_THREADS=512: two 77-KB workgroups per CU, 288-byte row segments)
#endif
#ifndef NTT28_THREADS
#define NTT28_THREADS 256
#endif
#define NTT28_TILE (1u << NTT28_TILE_LOG)
// Round 5: the passes multiply on NINE limbs of 29 bits (R' = 2^261: 162 + 9 mads per product instead of 200 + 10, nine-limb additions, 36-byte LDS elements).
// Six spare bits instead of 25: mul takes B(a) B(b) <= 70 (BLS12-381) / 169 (BN254), so the bound plan below does not apply -- every two-stage round enters with
// all elements below 6r, uses the CONSTANT biases K1 = 8r, K2 = 16r, K3 = 4r, and weakly reduces (wred, ~30 instructions) the outputs no multiplication touched:
// x0' = a0 + a1 (always) and, where the twiddle index is 0, x1', x2', x3' as well.  Every product operand then stays below 2 * 6 + 16 = 28 (static_assert below),
// every wred input below 28 < 128.  -DZL_NTT_FR28 keeps round 4's 10 x 28-bit instance (A/B).
template <class FrP> struct Fr28Of;
#ifdef ZL_NTT_FR28
template <> struct Fr28Of<BLS12_381_Fr> { using type = BLS12_381_Fr28; };
template <> struct Fr28Of<BN254_Fr> { using type = BN254_Fr28; };
#else
template <> struct Fr28Of<BLS12_381_Fr> { using type = BLS12_381_Fr29; };
template <> struct Fr28Of<BN254_Fr> { using type = BN254_Fr29; };
#endif
template <class P28> struct NttTight { static constexpr bool value = P28::MUL_BOUND < 1024; };  // few spare bits: constant biases + a weak reduction per round
static_assert(BLS12_381_Fr29::MUL_BOUND >= 28 && BN254_Fr29::MUL_BOUND >= 28 && BLS12_381_Fr29::KQ_MAX >= 4, "tight rounds: product operands reach 2 * 6 + 16 = 28 r");
// bias exponents per two-stage round of a pass (round 0 enters with B = 6: a caller's canonical-or-not 256-bit input is below 5.3 r, a product below 2 r)
struct Ntt28Plan {
    int j1[6], j2[6];              // K1 = 2^j1 r >= (B + 2) r, K2 = 2^j2 r >= (2 B + 2) r; the single last stage of an odd s uses K1 of its round index
    unsigned long long bound[7];   // B entering round r
};
constexpr Ntt28Plan ntt28_plan() {
    Ntt28Plan p{};
    unsigned long long B = 6;
    for (int r = 0; r < 6; r++) {
        p.bound[r] = B;
        int j1 = 0, j2 = 0;
        while ((1ull << j1) < B + 2) j1++;
        while ((1ull << j2) < 2 * B + 2) j2++;
        p.j1[r] = j1;
        p.j2[r] = j2;
        const unsigned long long b1 = 4 * B, b2 = B + (1ull << j1) + 4, b3 = 2 * B + (1ull << j2);
        B = b1 > b2 ? (b1 > b3 ? b1 : b3) : (b2 > b3 ? b2 : b3);
    }
    p.bound[6] = B;
    return p;
}
constexpr Ntt28Plan NTT28_PLAN = ntt28_plan();
static_assert(NTT28_PLAN.bound[5] < (1ull << 16) && NTT28_PLAN.j2[4] <= 20 && NTT28_PLAN.j1[5] <= 20, "lazy NTT bounds: five rounds (s <= 10) stay far below the multiplier's 2^25 and wred's 2^20");
// the exponents are linear in the round index, which is how the kernel computes them (a runtime index into a host constexpr object is not device code)
constexpr bool ntt28_plan_is_linear() {
    for (int r = 0; r < 6; r++)
        if (NTT28_PLAN.j1[r] != 3 + 2 * r || NTT28_PLAN.j2[r] != 4 + 2 * r) return false;
    return true;
}
static_assert(ntt28_plan_is_linear(), "k_ntt_pass28 computes j1 = 3 + 2 r, j2 = 4 + 2 r");
template <class P28>
__device__ __forceinline__ void load_bias28(uint32_t (&K)[P28::L], int j) {  // uniform: constant-memory lookups
#pragma unroll
    for (int i = 0; i < P28::L; i++) K[i] = P28::kq(j, i);
}
template <class E>
__device__ __forceinline__ E lds_load28(const uint32_t* sh, uint32_t pos) {
    E r;
#pragma unroll
    for (int k = 0; k < E::L; k++) r.l[k] = sh[k * NTT28_TILE + pos];
    return r;
}
template <class E>
__device__ __forceinline__ void lds_store28(uint32_t* sh, uint32_t pos, const E& v) {
#pragma unroll
    for (int k = 0; k < E::L; k++) sh[k * NTT28_TILE + pos] = v.l[k];
}
template <class P28>
__device__ __forceinline__ Fr28<P28> load28(const void* __restrict__ base, uint64_t idx) {  // one 32-byte element -> 10 limbs
    const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * idx;
    const uint4 lo = p[0], hi = p[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return zl::unpack28r<P28>(w);
}
// the scratch vector between two lazy passes holds the ten limbs as they are (40 bytes per element, carried, bound as the pass left it): no weak
// reduction, no packing on the way out and no unpacking on the way in (90 instructions per element and pass boundary)
// (round 5: nine-limb elements take 36-byte slots -- 10 % less traffic between passes; a slot is only 4-byte aligned, so it moves as nine dwords)
template <class P28> struct ScratchBytes { static constexpr size_t value = (size_t)4 * P28::L; };
template <class P28>
__device__ __forceinline__ Fr28<P28> load40(const void* __restrict__ base, uint64_t idx) {
    Fr28<P28> r;
    if constexpr (P28::L & 1) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(base) + (size_t)P28::L * idx;
#pragma unroll
        for (int k = 0; k < P28::L; k++) r.l[k] = p[k];
    } else {
        const uint2* p = reinterpret_cast<const uint2*>(base) + (P28::L / 2) * idx;
#pragma unroll
        for (int k = 0; k < P28::L / 2; k++) {
            const uint2 v = p[k];
            r.l[2 * k] = v.x;
            r.l[2 * k + 1] = v.y;
        }
    }
    return r;
}
template <class P28>
__device__ __forceinline__ void store40(void* __restrict__ base, uint64_t idx, const Fr28<P28>& x) {
    if constexpr (P28::L & 1) {
        uint32_t* p = reinterpret_cast<uint32_t*>(base) + (size_t)P28::L * idx;
#pragma unroll
        for (int k = 0; k < P28::L; k++) p[k] = x.l[k];
    } else {
        uint2* p = reinterpret_cast<uint2*>(base) + (P28::L / 2) * idx;
#pragma unroll
        for (int k = 0; k < P28::L / 2; k++) p[k] = make_uint2(x.l[2 * k], x.l[2 * k + 1]);
    }
}
template <class P28>
__device__ __forceinline__ Fr28<P28> twiddle2_28(const void* lo, const void* hi, uint32_t L, uint64_t e) {
    return zl::mul(load28<P28>(lo, e & ((1ull << L) - 1)), load28<P28>(hi, e >> L));
}
template <class P28, class WordsFn>
__device__ __forceinline__ Fr28<P28> const28(WordsFn f) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = f(i);
    return zl::unpack28r<P28>(w);
}

// Two register budgets of the same kernel (round 5): MINB = 2 lets the compiler take the 101-104 registers the nine-limb pass wants (a transform on its own:
// 2.13 ms at 2^24); MINB = 5 caps it at 96 (6 spilled dwords, 2.24 ms), which is what fits beside the 416-register G2 accumulation of a Groth16 proof -- the
// witness map of a proof runs there (zl_ctx::ntt_fit_beside; uncapped passes waited 5 ms for the accumulation to end, profiles/r05_g16_ntt_regs_ab.log)
template <class FrP, bool LAST, int MINB = 2>
__global__ void __launch_bounds__(NTT28_THREADS, MINB) k_ntt_pass28(const Fp<FrP>* __restrict__ in, Fp<FrP>* __restrict__ out, NttArgs a) {
    using P28 = typename Fr28Of<FrP>::type;
    in = reinterpret_cast<const Fp<FrP>*>(reinterpret_cast<const unsigned char*>(in) + (size_t)blockIdx.y * a.in_batch);
    out = reinterpret_cast<Fp<FrP>*>(reinterpret_cast<unsigned char*>(out) + (size_t)blockIdx.y * a.out_batch);
    using E = Fr28<P28>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* sh = reinterpret_cast<uint32_t*>(smem);                          // [L][NTT28_TILE]
    const uint32_t tid = threadIdx.x;
    const uint32_t s = a.s, logC = a.logC, R = 1u << s, C = 1u << logC;
    E* sh_w = reinterpret_cast<E*>(smem + (size_t)E::L * 4 * NTT28_TILE);       // [2^(s-1)] butterfly roots
    constexpr bool TIGHT = NttTight<P28>::value;
    E* sh_row = sh_w + (a.w_unpacked ? 0u : (R >> 1));                          // [2^s] per-row twiddles (middle passes without a row table)
    const uint32_t n_log = a.n_log;

    // ---- tile addressing (as k_ntt_pass) --------------------------------------------------------------------
    uint64_t in_base, out_base, in_row, in_col, out_row;
    uint64_t K0 = 0, tile_hi = 0;
    {
        const uint64_t tile = blockIdx.x;
        if (!LAST) {
            const uint32_t stride_log = n_log - a.S_prev - s;
            const uint64_t lo_blocks = (1ull << stride_log) >> logC;
            const uint64_t hi = tile / lo_blocks, lo0 = (tile % lo_blocks) << logC;
            tile_hi = hi;
            in_base = (hi << (s + stride_log)) + lo0;
            out_base = in_base;
            in_row = 1ull << stride_log;
            in_col = 1;
            out_row = in_row;
            uint64_t rem = hi;
            uint32_t Sq = a.S_prev;
            for (int q = (int)a.p - 2; q >= 0; q--) {
                Sq -= a.sizes[q];
                K0 += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
                rem >>= a.sizes[q];
            }
        } else if (a.P == 1) {
            in_base = out_base = 0;
            in_row = out_row = 1;
            in_col = 1;
        } else {
            const uint32_t s1 = a.sizes[0];
            const uint32_t rest_log = a.S_prev - s1;
            const uint64_t k1_blocks = (1ull << s1) >> logC;
            const uint64_t k1_0 = (tile % k1_blocks) << logC, rest = tile / k1_blocks;
            in_base = ((k1_0 << rest_log) + rest) << s;
            in_col = 1ull << (rest_log + s);
            in_row = 1;
            uint64_t rem = rest, Krest = 0;
            uint32_t Sq = a.S_prev;
            for (int q = (int)a.P - 2; q >= 1; q--) {
                Sq -= a.sizes[q];
                Krest += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
                rem >>= a.sizes[q];
            }
            K0 = k1_0 + Krest;
            out_base = K0;
            out_row = 1ull << a.S_prev;
        }
    }
    // ---- stage the small tables (unpacked once per tile) ----------------------------------------------------
    {
        if (!a.w_unpacked)
            for (uint32_t i = tid; i < (R >> 1); i += NTT28_THREADS) sh_w[i] = load28<P28>(a.w_small, i);
        if (!LAST && a.p > 1) {
            // the row twiddles depend on the tile's high index only: tabulated once per (size, direction) (k_ntt_row_table28) -- a tile of 1024 elements
            // would otherwise pay 2^s products for them (measured: the middle pass of 2^24 0.92 ms against 0.85 for the 32-bit kernel's 2048-element tiles)
            if (!a.row_tw) {
                const uint32_t shift = n_log - a.S_prev - s;
                for (uint32_t r = tid; r < R; r += NTT28_THREADS) sh_row[r] = twiddle2_28<P28>(a.t_lo, a.t_hi, a.L, ((uint64_t)r * K0) << shift);
            }
        }
    }
    __syncthreads();
    auto load_elem = [&](uint32_t r, uint32_t col) -> E {
        const uint64_t m = in_base + (uint64_t)r * in_row + (uint64_t)col * in_col;
        E x = a.p == 1 ? load28<P28>(in, m) : load40<P28>(in, m);
        if (a.p == 1) {
            if (a.to_mont) x = zl::mul(x, const28<P28>([](int i) { return P28::to_mont(i); }));
            if (a.pre_coset) x = zl::mul(x, twiddle2_28<P28>(a.g_lo, a.g_hi, a.L, m));
        } else if (!LAST) {
            // (the tabulated row twiddles come straight from global memory: 8 KB per high index, shared by all its tiles and cache-resident -- the tile
            // then needs no LDS for them and a middle pass fits three workgroups per CU like the others)
            x = zl::mul(x, a.row_tw ? load28<P28>(a.row_tw, (tile_hi << s) + r) : sh_row[r]);
        } else if (a.last_tw) {
            x = zl::mul(x, load28<P28>(a.last_tw, m));
        } else {
            x = zl::mul(x, twiddle2_28<P28>(a.t_lo, a.t_hi, a.L, (uint64_t)r * (K0 + col)));
        }
        return x;
    };
    auto store_elem = [&](uint32_t k, uint32_t col, E x) {
        const uint64_t m = out_base + (uint64_t)k * out_row + col;
        if (LAST) {
            if (a.post_coset) {
                x = zl::mul(x, twiddle2_28<P28>(a.g_lo, a.g_hi, a.L, m));
            } else if (a.post_scale) {
                uint32_t w[8];
#pragma unroll
                for (int i = 0; i < 8; i++) w[i] = a.ninv[i];
                x = zl::mul(x, zl::unpack28r<P28>(w));
            }
            if (a.from_mont) x = zl::mul(x, const28<P28>([](int i) { return P28::from_mont(i); }));
            x = zl::canon(x);
        } else {
            store40<P28>(out, m, x);
            return;
        }
        uint32_t w[8];
        zl::pack28r<P28>(w, x);
        uint4* q = reinterpret_cast<uint4*>(out) + 2 * m;
        q[0] = make_uint4(w[0], w[1], w[2], w[3]);
        q[1] = make_uint4(w[4], w[5], w[6], w[7]);
    };
    // butterfly roots: from LDS (staged above), or -- experiment -- straight from a limb-form table in global memory (cache-resident: 2^(s-1) x 36 B), which
    // frees 4.6 KB of LDS per workgroup: 36.9 KB instead of 41.5, i.e. four workgroups per CU instead of three
    const E* w_glob = reinterpret_cast<const E*>(a.w_unpacked);
    auto root = [&](uint32_t i) -> E { return w_glob ? w_glob[i] : sh_w[i]; };
    uint32_t K1[E::L], K2[E::L], K3[E::L];
    load_bias28<P28>(K3, 2);  // 4 r: a2 - a3 (a3 is always a product, < 2r; a2 a product or, at k2 = 0 in a tight round, below 6 + 8 -- then nothing multiplies the difference)
    if (TIGHT) {              // constant biases: every round enters below 6 r
        load_bias28<P28>(K1, 3);
        load_bias28<P28>(K2, 4);
    }
    auto quad = [&](E& x0, E& x1, E& x2, E& x3, uint32_t k2, uint32_t hl) {
        const uint32_t h2 = 1u << hl, sh1 = s - 2 - hl;
        // carry passes only where a value becomes a subtrahend or goes back to LDS (3 per quad instead of 8): everything else feeds a product or the
        // minuend side of a biased subtraction with fat limbs (< 2^31; zl_field28r.h)
        const E a0 = zl::add_nc(x0, x2), a1 = zl::add(x1, x3);
        E a2 = zl::subk_nc(x0, x2, K1);
        if (k2 != 0) a2 = zl::mul(a2, root(k2 << sh1));
        const E a3 = zl::mul(zl::subk_nc(x1, x3, K1), root((k2 + h2) << sh1));
        x0 = zl::add(a0, a1);
        x2 = zl::add(a2, a3);
        x1 = zl::subk_nc(a0, a1, K2);
        x3 = zl::subk_nc(a2, a3, K3);
        if (TIGHT) x0 = zl::wred(x0);  // 4 B <= 24 -> < 2r: the one output of a quad that no product ever touches
        if (k2 != 0) {
            const E w = root(k2 << (sh1 + 1));
            x1 = zl::mul(x1, w);
            x3 = zl::mul(x3, w);
        } else {
            zl::carry28r(x1);
            zl::carry28r(x3);
            if (TIGHT) {  // trivial twiddles: B + 8 + 2, 2 B + 16, B + 8 + 4 <= 28 -> < 2r
                x1 = zl::wred(x1);
                x2 = zl::wred(x2);
                x3 = zl::wred(x3);
            }
        }
    };
    for (uint32_t idx = tid; idx < R * C; idx += NTT28_THREADS) {
        const uint32_t col = idx & (C - 1), r = idx >> logC;
        lds_store28(sh, tile_pos(r, col, logC), load_elem(r, col));
    }
    __syncthreads();
    uint32_t hl = s, rd = 0;
    while (hl >= 2u) {
        hl -= 2;
        const uint32_t h2 = 1u << hl, h = h2 << 1;
        if (!TIGHT) {
            load_bias28<P28>(K1, 3 + 2 * (int)rd);
            load_bias28<P28>(K2, 4 + 2 * (int)rd);
        }
        rd++;
        for (uint32_t q = tid; q < (R >> 2) * C; q += NTT28_THREADS) {
            const uint32_t col = q & (C - 1), rr = q >> logC;
            const uint32_t k2 = rr & (h2 - 1), blk = rr >> hl;
            const uint32_t i0 = (blk << (hl + 2)) | k2;
            const uint32_t p0 = tile_pos(i0, col, logC), p1 = tile_pos(i0 + h2, col, logC), p2 = tile_pos(i0 + h, col, logC),
                           p3 = tile_pos(i0 + h + h2, col, logC);
            E x0 = lds_load28<E>(sh, p0), x1 = lds_load28<E>(sh, p1), x2 = lds_load28<E>(sh, p2), x3 = lds_load28<E>(sh, p3);
            quad(x0, x1, x2, x3, k2, hl);
            lds_store28(sh, p0, x0);
            lds_store28(sh, p1, x1);
            lds_store28(sh, p2, x2);
            lds_store28(sh, p3, x3);
        }
        __syncthreads();
    }
    if (hl == 1) {  // odd s: the last stage (distance 1) on its own (tight: enters below 4 r, leaves below 4 + 8)
        if (!TIGHT) load_bias28<P28>(K1, 3 + 2 * (int)rd);
        for (uint32_t q = tid; q < (R >> 1) * C; q += NTT28_THREADS) {
            const uint32_t col = q & (C - 1), rr = q >> logC;
            const uint32_t i = rr << 1;
            const uint32_t pu = tile_pos(i, col, logC), pv = tile_pos(i + 1, col, logC);
            const E u = lds_load28<E>(sh, pu), v = lds_load28<E>(sh, pv);
            lds_store28(sh, pu, zl::add(u, v));
            lds_store28(sh, pv, zl::subk(u, v, K1));
        }
        __syncthreads();
    }
    for (uint32_t idx = tid; idx < R * C; idx += NTT28_THREADS) {
        const uint32_t col = idx & (C - 1), k = idx >> logC;
        const uint32_t row = s ? (__brev(k) >> (32 - s)) : 0;
        store_elem(k, col, lds_load28<E>(sh, tile_pos(row, col, logC)));
    }
}
// tables of the lazy passes: every entry x R (32-bit Montgomery form, as the table kernels write it) becomes the canonical integer x R' mod r
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_to_lazy(Fp<FrP>* __restrict__ table, size_t count) {
    using F = Fp<FrP>;
    using P28 = typename Fr28Of<FrP>::type;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    F rp;
#pragma unroll
    for (int k = 0; k < F::N; k++) rp.l[k] = P28::rp(k);
    table[i] = zl::mul(table[i], rp);  // (x R)(R') / R = x R', reduced below r
}
// table[i] (32-byte canonical words) -> its limbs (4 L bytes), for the passes that read butterfly roots from global memory
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_unpack_table28(const Fp<FrP>* __restrict__ table, uint32_t* __restrict__ out, uint32_t count) {
    using P28 = typename Fr28Of<FrP>::type;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const Fr28<P28> e = load28<P28>(table, i);
#pragma unroll
    for (int k = 0; k < P28::L; k++) out[(size_t)P28::L * i + k] = e.l[k];
}
// the combined twiddles of the last pass (k_ntt_last_table) in the lazy multiplier's form
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_last_table28(Fp<FrP>* __restrict__ table, NttArgs a) {
    using P28 = typename Fr28Of<FrP>::type;
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >> a.n_log) return;
    const uint32_t s = a.s, s1 = a.sizes[0], rest_log = a.S_prev - s1;
    const uint64_t r = m & ((1ull << s) - 1), rest = (m >> s) & ((1ull << rest_log) - 1), k1 = m >> (s + rest_log);
    uint64_t rem = rest, Krest = 0;
    uint32_t Sq = a.S_prev;
    for (int q = (int)a.P - 2; q >= 1; q--) {
        Sq -= a.sizes[q];
        Krest += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
        rem >>= a.sizes[q];
    }
    const Fr28<P28> t = zl::canon(twiddle2_28<P28>(a.t_lo, a.t_hi, a.L, r * (k1 + Krest)));
    uint32_t w[8];
    zl::pack28r<P28>(w, t);
    uint4* q = reinterpret_cast<uint4*>(table) + 2 * m;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// the per-row twiddles of a middle pass for every value of the tile's high index: table[(hi << s) + r] = w_N^((r K0(hi)) << shift), K0 as in k_ntt_pass28
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_row_table28(Fp<FrP>* __restrict__ table, NttArgs a) {
    using P28 = typename Fr28Of<FrP>::type;
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >> (a.S_prev + a.s)) return;
    const uint64_t r = m & ((1ull << a.s) - 1), hi = m >> a.s;
    uint64_t rem = hi, K0 = 0;
    uint32_t Sq = a.S_prev;
    for (int q = (int)a.p - 2; q >= 0; q--) {
        Sq -= a.sizes[q];
        K0 += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
        rem >>= a.sizes[q];
    }
    const uint32_t shift = a.n_log - a.S_prev - a.s;
    const Fr28<P28> t = zl::canon(twiddle2_28<P28>(a.t_lo, a.t_hi, a.L, (r * K0) << shift));
    uint32_t w[8];
    zl::pack28r<P28>(w, t);
    uint4* q = reinterpret_cast<uint4*>(table) + 2 * m;
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// table[i] = scale * base^(i << shift), i < count
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_pow_table(Fp<FrP>* __restrict__ table, uint32_t count, uint32_t shift, Fp<FrP> base, Fp<FrP> scale) {
    using F = Fp<FrP>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t e = (uint64_t)i << shift;
    F acc = scale, b = base;
    while (e) {
        if (e & 1) acc = zl::mul(acc, b);
        b = zl::sqr(b);
        e >>= 1;
    }
    table[i] = acc;
}

// The last pass multiplies element (row r, digit-reversed column index K) by w_N^(r K): 2^n distinct factors, so any split into small
// tables costs a second multiplication per element to combine them (15 instead of 14 multiplications per element at 2^24, and the passes
// are bound by exactly that count: 0.68 / 0.85 / 1.00 ms for 4 / 5 / 6 multiplications per element).  HBM is the resource with slack
// (0.43 of 8 TB/s), so the combined factors are tabulated once per (size, direction) in the order the last pass loads its input and
// streamed beside it: +32 B of reads per element for one multiplication less.  table[m] = scale * w^(r (k1 + Krest)) with
// m = ((k1 << rest_log) + rest) << s | r, exactly the addressing of k_ntt_pass<LAST>.
template <class FrP>
__global__ void __launch_bounds__(256) k_ntt_last_table(Fp<FrP>* __restrict__ table, NttArgs a) {
    using F = Fp<FrP>;
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >> a.n_log) return;
    const uint32_t s = a.s, s1 = a.sizes[0], rest_log = a.S_prev - s1;
    const uint64_t r = m & ((1ull << s) - 1), rest = (m >> s) & ((1ull << rest_log) - 1), k1 = m >> (s + rest_log);
    uint64_t rem = rest, Krest = 0;
    uint32_t Sq = a.S_prev;
    for (int q = (int)a.P - 2; q >= 1; q--) {
        Sq -= a.sizes[q];
        Krest += (rem & ((1ull << a.sizes[q]) - 1)) << Sq;
        rem >>= a.sizes[q];
    }
    table[m] = twiddle2(reinterpret_cast<const F*>(a.t_lo), reinterpret_cast<const F*>(a.t_hi), a.L, r * (k1 + Krest));
}

// ------------------------------------------------------------------------------------------------ host driver
struct NttPlan {
    uint32_t P;
    uint32_t sizes[4];
};
static NttPlan ntt_plan(unsigned n) {
    NttPlan pl{};
    if (n <= NTT_MAX_S) { pl.P = 1; pl.sizes[0] = n; return pl; }
    uint32_t P = (n + 7) / 8;
    if (P > 4) P = 4;
    pl.P = P;
    // even factor sizes where possible (every pass then runs whole two-stage rounds, first and last fused with the global accesses):
    // start from the largest even size <= n / P and hand the remainder out in steps of two, a final odd bit to the last factor
    uint32_t base = (n / P) & ~1u, rem = n - base * P;
    for (uint32_t p = 0; p < P; p++) pl.sizes[p] = base;
    for (uint32_t p = 0; p < P && rem >= 2 && pl.sizes[p] + 2 <= NTT_MAX_S; p++) { pl.sizes[p] += 2; rem -= 2; }
    for (uint32_t p = P; p-- > 0 && rem > 0;) {
        const uint32_t room = NTT_MAX_S - pl.sizes[p], add = rem < room ? rem : room;
        pl.sizes[p] += add;
        rem -= add;
    }
    return pl;
}

// lazy: the tables of the 28-bit passes (entries w R' mod r as canonical integers) -- a separate set: the distributed transform's cross kernel
// (k_ntt_cross) keeps the 32-bit field and its Montgomery tables
template <class FrP>
static int ntt_tables(zl_ctx* ctx, int curve, unsigned n, bool inverse, zl_twiddles** out, bool lazy = false) {
    using F = Fp<FrP>;
    const uint64_t key = ((uint64_t)curve << 16) | (lazy ? 0x100u : 0u) | ((uint64_t)n << 1) | (inverse ? 1 : 0);
    auto it = ctx->twiddles.find(key);
    if (it != ctx->twiddles.end()) { *out = &it->second; return ZL_OK; }
    // root of unity of order 2^n (host)
    F w;
    for (int i = 0; i < F::N; i++) w.l[i] = FrP::two_adic_root(i);
    for (unsigned i = n; i < (unsigned)FrP::TWO_ADICITY; i++) w = zl::sqr(w);
    F g;
    for (int i = 0; i < F::N; i++) g.l[i] = FrP::generator(i);
    F scale = F::one();
    if (inverse) {
        w = zl::inv(w);
        g = zl::inv(g);
        scale = zl::inv(zl::from_u64<FrP>(1ull << n));
    }
    zl_twiddles tw;
    const unsigned L = (n + 1) / 2;
    tw.lo_bits = L;
    const uint32_t n_lo = 1u << L, n_hi = 1u << (n - L);
    // layout: t_lo | t_hi | g_lo | g_hi | small tables for s = 1..NTT_MAX_S (2^(s-1) each) | inverse only: n^-1 * t_hi (the last pass of a
    // multi-pass inverse transform takes its inter-factor twiddles from it, so the 1/n scaling costs no multiplication of its own)
    const size_t small_total = (1u << NTT_MAX_S);
    const size_t total = (size_t)2 * (n_lo + n_hi) + small_total + (inverse ? n_hi : 0);
    F* d;
    ZL_HIP(ctx, hipMalloc((void**)&d, total * sizeof(F)));
    hipStream_t st = ctx->stream;
    F* t_lo = d;
    F* t_hi = t_lo + n_lo;
    F* g_lo = t_hi + n_hi;
    F* g_hi = g_lo + n_lo;
    F* small = g_hi + n_hi;
    const F one = F::one();
    hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((n_lo + 255) / 256), dim3(256), 0, st, t_lo, n_lo, 0u, w, one);
    hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((n_hi + 255) / 256), dim3(256), 0, st, t_hi, n_hi, L, w, one);
    hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((n_lo + 255) / 256), dim3(256), 0, st, g_lo, n_lo, 0u, g, one);
    hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((n_hi + 255) / 256), dim3(256), 0, st, g_hi, n_hi, L, g, scale);
    if (inverse) hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((n_hi + 255) / 256), dim3(256), 0, st, small + small_total, n_hi, L, w, scale);
    // small[s] at offset 2^(s-1): w_(2^s)^i = w^(i << (n - s)), i < 2^(s-1)
    for (unsigned s = 1; s <= NTT_MAX_S && s <= n; s++) {
        const uint32_t cnt = 1u << (s - 1);
        hipLaunchKernelGGL((k_ntt_pow_table<FrP>), dim3((cnt + 255) / 256), dim3(256), 0, st, small + cnt, cnt, n - s, w, one);
    }
    if (lazy) hipLaunchKernelGGL((k_ntt_to_lazy<FrP>), dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, st, d, total);
    ZL_HIP(ctx, hipGetLastError());
    tw.d_lo = t_lo;
    tw.d_hi = t_hi;
    tw.d_small = small;
    ctx->twiddles[key] = tw;
    *out = &ctx->twiddles[key];
    return ZL_OK;
}

template <class FrP>
// count > 1: `count` equal-size transforms with the same flags, vector v at d_data + v * stride_bytes, in ONE launch per pass (grid.y = count)
static int ntt_run_t(zl_ctx* ctx, int curve, void* d_data, unsigned n, unsigned flags, unsigned count = 1, size_t stride_bytes = 0) {
    using F = Fp<FrP>;
    if (n > (unsigned)FrP::TWO_ADICITY || n > 30) return ZL_EINVAL;
    const bool inverse = flags & ZL_INVERSE, coset = flags & ZL_COSET;
    const bool mont_in = flags & (ZL_MONT | ZL_MONT_IN), mont_out = flags & (ZL_MONT | ZL_MONT_OUT);
    ctx->timing = zl_timing{};
    if (n == 0) {
        // one element: forward/inverse are the identity (n^-1 = 1, g^0 = 1)
        if (mont_in != mont_out) return ZL_EINVAL;
        return ZL_OK;
    }
    // Round 4: the passes run on lazily reduced 28-bit limbs (k_ntt_pass28) unless ZL_NTT_NO_LAZY is set (developer A/B switch; the 32-bit passes stay)
    static const bool lazy = getenv("ZL_NTT_NO_LAZY") == nullptr;
    const bool roots_global = true;  // round 5: the butterfly roots as limbs from global memory (2^24: 2.16 -> 2.13 ms, profiles/r05_ntt_fr29_ab2.log); the LDS-staged path stays in the kernel for w_unpacked == nullptr
    zl_twiddles* tw;
    int rc;
    if ((rc = ntt_tables<FrP>(ctx, curve, n, inverse, &tw, lazy))) return rc;
    const NttPlan pl = ntt_plan(n);
    const size_t N = (size_t)1 << n;
    F* data = reinterpret_cast<F*>(d_data);
    F* scratch = nullptr;
    if (pl.P > 1) {
        void* p;
        if ((rc = zl_scratch_get(ctx, 6, (size_t)count * N * (lazy ? ScratchBytes<typename Fr28Of<FrP>::type>::value : sizeof(F)), &p))) return rc;  // lazy passes: the limbs as they are between passes
        scratch = reinterpret_cast<F*>(p);
    }
    const unsigned L = tw->lo_bits;
    const F* t_lo = reinterpret_cast<const F*>(tw->d_lo);
    const F* t_hi = reinterpret_cast<const F*>(tw->d_hi);
    const F* g_lo = t_hi + ((size_t)1 << (n - L));
    const F* g_hi = g_lo + ((size_t)1 << L);
    const F* small = reinterpret_cast<const F*>(tw->d_small);
    F ninv = zl::inv(zl::from_u64<FrP>(1ull << n));
    if (lazy) {  // n^-1 R -> n^-1 R' (canonical), the multiplier form of the lazy passes
        using P28 = typename Fr28Of<FrP>::type;
        F rp;
        for (int k = 0; k < F::N; k++) rp.l[k] = P28::rp(k);
        ninv = zl::mul(ninv, rp);
    }
    hipStream_t st = ctx->stream;
    // tiles + tables can exceed the 64 KiB default dynamic-LDS limit (gfx950 has 160 KiB per CU); per device, so set per call
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass<FrP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass<FrP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass28<FrP, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass28<FrP, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass28<FrP, true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ntt_pass28<FrP, false, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (ctx->timing_on) ZL_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    uint32_t S_prev = 0;
    // byte budget of the per-(size, direction) tables of this ctx -- the last pass's combined twiddles (N x 32 B) and the middle passes' row tables (<= 32 MB each):
    // least recently used keys go first, with ALL their tables; the stream is drained before a table that may be in flight is freed
    const size_t table_budget = (size_t)zl_tune("ZL_TUNE_NTT_LAST_MB", 2048) << 20;  // 2^24 forward + inverse = 1 GiB + row tables; a sweep over sizes stays bounded
    int rc_t = ZL_OK;
    auto make_room = [&](size_t want) -> int {
        while (ctx->ntt_last_bytes + want > table_budget) {
            zl_twiddles* victim = nullptr;
            for (auto& kv : ctx->twiddles)
                if (kv.second.last_bytes && &kv.second != tw && (!victim || kv.second.last_used < victim->last_used)) victim = &kv.second;
            if (!victim) break;
            ZL_HIP(ctx, hipStreamSynchronize(st));
            if (victim->d_last) (void)hipFree(victim->d_last);
            victim->d_last = nullptr;
            for (auto& r : victim->d_row) { if (r) (void)hipFree(r); r = nullptr; }
            ctx->ntt_last_bytes -= victim->last_bytes;
            victim->last_bytes = 0;
        }
        return ZL_OK;
    };
    for (uint32_t p = 1; p <= pl.P; p++) {
        const bool last = p == pl.P;
        NttArgs a{};
        a.n_log = n;
        a.s = pl.sizes[p - 1];
        a.P = pl.P;
        a.p = p;
        for (int k = 0; k < 4; k++) a.sizes[k] = pl.sizes[k];
        a.S_prev = S_prev;
        a.L = L;
        a.t_lo = t_lo;
        a.t_hi = t_hi;
        a.g_lo = g_lo;
        a.g_hi = g_hi;
        a.w_small = small + ((size_t)1 << (a.s ? a.s - 1 : 0));
        a.w_unpacked = nullptr;
        if (lazy && roots_global) {
            using P28u = typename Fr28Of<FrP>::type;
            if (!tw->d_small_limbs) {
                void* t = nullptr;
                if (hipMalloc(&t, (size_t)4 * P28u::L << NTT_MAX_S) == hipSuccess) {
                    hipLaunchKernelGGL((k_ntt_unpack_table28<FrP>), dim3((1u << NTT_MAX_S) / 256), dim3(256), 0, st, small, (uint32_t*)t, 1u << NTT_MAX_S);
                    tw->d_small_limbs = t;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (tw->d_small_limbs) a.w_unpacked = reinterpret_cast<const uint32_t*>(tw->d_small_limbs) + (size_t)P28u::L * ((size_t)1 << (a.s ? a.s - 1 : 0));
        }
        a.to_mont = (p == 1 && !mont_in) ? 1 : 0;
        if (lazy) a.to_mont = (p == 1 && !mont_in && mont_out) ? 1 : 0;  // the data keeps its form: a conversion only when the caller asks for one
        a.pre_coset = (p == 1 && coset && !inverse) ? 1 : 0;
        a.from_mont = (last && !mont_out) ? 1 : 0;
        if (lazy) a.from_mont = (last && mont_in && !mont_out) ? 1 : 0;
        a.post_coset = (last && coset && inverse) ? 1 : 0;
        a.post_scale = (last && inverse && !coset) ? 1 : 0;
        if (a.post_scale && pl.P > 1) {  // n^-1 rides on the last pass's twiddles (ntt_tables)
            a.t_hi = small + ((size_t)1 << NTT_MAX_S);
            a.post_scale = 0;
        }
        for (int k = 0; k < 8; k++) a.ninv[k] = ninv.l[k];
        // the combined twiddles of the last pass, tabulated once per (size, direction): forward (plain and coset share it: the coset scaling
        // rides on pass 1) and the scaled inverse; the coset inverse (its 1/n lives in the coset table) combines on the fly
        if (last && pl.P > 1 && !(inverse && coset) && n <= 26 && !getenv("ZL_NTT_NO_LAST_TABLE")) {
            const size_t want = N * sizeof(F);
            if (!tw->d_last && want <= table_budget) {
                if ((rc_t = make_room(want))) return rc_t;
                void* t = nullptr;
                if (ctx->ntt_last_bytes + want <= table_budget && hipMalloc(&t, want) == hipSuccess) {
                    a.last_tw = nullptr;
                    if (lazy) hipLaunchKernelGGL((k_ntt_last_table28<FrP>), dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, reinterpret_cast<F*>(t), a);
                    else hipLaunchKernelGGL((k_ntt_last_table<FrP>), dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, reinterpret_cast<F*>(t), a);
                    tw->d_last = t;
                    tw->last_bytes += want;
                    ctx->ntt_last_bytes += want;
                } else {
                    (void)hipGetLastError();  // no room: keep combining on the fly
                }
            }
            if (tw->d_last) tw->last_used = ++ctx->ntt_clock;
            a.last_tw = tw->d_last;
        }
        if (lazy && !last && p > 1 && p <= 4 && S_prev + a.s <= 20) {  // middle pass: its row twiddles per tile high index, <= 32 MB, kept with the tables
            if (!tw->d_row[p - 1]) {
                //  counted in the same byte budget as the last-pass tables and evicted with them: a sweep over sizes, directions and curves
                // would otherwise keep up to three 32-MB tables per key for ever
                const size_t want = (sizeof(F)) << (S_prev + a.s);
                void* t = nullptr;
                if ((rc_t = make_room(want))) return rc_t;
                if (ctx->ntt_last_bytes + want <= table_budget && hipMalloc(&t, want) == hipSuccess) {
                    const uint64_t cnt = 1ull << (S_prev + a.s);
                    hipLaunchKernelGGL((k_ntt_row_table28<FrP>), dim3((uint32_t)((cnt + 255) / 256)), dim3(256), 0, st, reinterpret_cast<F*>(t), a);
                    tw->d_row[p - 1] = t;
                    tw->last_bytes += want;
                    ctx->ntt_last_bytes += want;
                } else {
                    (void)hipGetLastError();  // no room: the pass forms its row twiddles per tile (the LDS size below accounts for it)
                }
            }
            if (tw->d_row[p - 1]) tw->last_used = ++ctx->ntt_clock;
            a.row_tw = tw->d_row[p - 1];
        }
        // columns per tile
        uint32_t cols_avail_log;
        if (pl.P == 1) cols_avail_log = 0;
        else if (!last) cols_avail_log = n - S_prev - a.s;
        else cols_avail_log = pl.sizes[0];
        uint32_t logC = (lazy ? (uint32_t)NTT28_TILE_LOG : 11u) - a.s;  // NTT_TILE = 2^11, NTT28_TILE = 2^NTT28_TILE_LOG
        if (a.s > 10) return ZL_EINVAL;
        if (logC > cols_avail_log) logC = cols_avail_log;
        a.logC = logC;
        const uint64_t tiles = (uint64_t)N >> (a.s + logC);
        const F* src = (p == 1) ? data : scratch;
        F* dst = last ? data : scratch;
        const size_t scratch_vec = N * (lazy ? ScratchBytes<typename Fr28Of<FrP>::type>::value : sizeof(F));
        a.in_batch = (p == 1) ? stride_bytes : scratch_vec;
        a.out_batch = last ? stride_bytes : scratch_vec;
        if (lazy) {
            // tile + butterfly roots (+ the per-row twiddles of a middle pass that has no row table): 45 KB at s = 8 -> three workgroups per CU
            using E28 = Fr28<typename Fr28Of<FrP>::type>;
            const size_t lds28 = (size_t)E28::L * 4 * NTT28_TILE + sizeof(E28) * ((a.w_unpacked ? 0 : ((size_t)1 << a.s) / 2) + ((!last && p > 1 && !a.row_tw) ? ((size_t)1 << a.s) : 0));
            if (ctx->ntt_fit_beside && NTT28_THREADS == 256) {
                if (last) hipLaunchKernelGGL((k_ntt_pass28<FrP, true, 5>), dim3((uint32_t)tiles, count), dim3(NTT28_THREADS), lds28, st, src, dst, a);
                else hipLaunchKernelGGL((k_ntt_pass28<FrP, false, 5>), dim3((uint32_t)tiles, count), dim3(NTT28_THREADS), lds28, st, src, dst, a);
            } else {
                if (last) hipLaunchKernelGGL((k_ntt_pass28<FrP, true, 2>), dim3((uint32_t)tiles, count), dim3(NTT28_THREADS), lds28, st, src, dst, a);
                else hipLaunchKernelGGL((k_ntt_pass28<FrP, false, 2>), dim3((uint32_t)tiles, count), dim3(NTT28_THREADS), lds28, st, src, dst, a);
            }
        } else {
        const size_t lds = (size_t)F::N * 4 * NTT_TILE + sizeof(F) * (((size_t)1 << a.s) / 2 + ((size_t)1 << a.s));
        if (last) hipLaunchKernelGGL((k_ntt_pass<FrP, true>), dim3((uint32_t)tiles, count), dim3(NTT_THREADS), lds, st, src, dst, a);
        else hipLaunchKernelGGL((k_ntt_pass<FrP, false>), dim3((uint32_t)tiles, count), dim3(NTT_THREADS), lds, st, src, dst, a);
        }
        S_prev += a.s;
    }
    ZL_HIP(ctx, hipGetLastError());
    if (ctx->timing_on) {
        ZL_HIP(ctx, hipEventRecord(ctx->ev[1], st));
        ZL_HIP(ctx, hipStreamSynchronize(st));
        ZL_HIP(ctx, hipEventElapsedTime(&ctx->timing.total_ms, ctx->ev[0], ctx->ev[1]));
        ctx->timing.dominant_ms = ctx->timing.total_ms;
    }
    ctx->timing.launches = pl.P;
    return ZL_OK;
}

// ------------------------------------------------------------------------------------------------ distributed transform
// One rank's "cross" step of a 2^n-point transform spread over G = 2^lg ranks (four-step with ONE
// all-to-all).  M = N/G, B = M/G.  Index split  j = j1*M + j2 (j1 < G),  k = k1 + G*k2 (k1 < G):
//     X[k1 + G k2] = sum_j2 w_M^(j2 k2) * [ w_N^(j2 k1) * sum_j1 x[j1 M + j2] w_G^(j1 k1) ]
// "block-column" layout: rank g holds x[j1*M + g*B + c] at local [j1*B + c];  "cyclic" layout: rank k1 holds X[k1 + G*k2] at
// local [k2].  Forward: this kernel on block-column data (G-point DFT down each local column, then the twiddle), all-to-all
// (row k1 -> rank k1), local M-point transform.  Inverse: local inverse M-point transform on cyclic data, all-to-all, this
// kernel (twiddle, inverse G-point DFT, 1/G).  Coset factors h^(j1 M + j2) split into a per-row constant h^(M j1) and a
// per-column factor h^j2 that rides on the twiddle.  One thread owns one column: G loads, G stores, all arithmetic in registers.
struct CrossArgs {
    uint32_t n_log, lg, logB, L, rank;
    uint32_t inverse, coset, to_mont, from_mont;
    const void *t_lo, *t_hi, *g_lo, *g_hi, *w_small;  // tables of the full 2^n domain (inverse tables when inverse)
    uint32_t cst[8];           // constant folded into the twiddle: 1 (forward), 1/G (inverse), M (inverse coset: g_hi carries 1/N)
    uint32_t row[16][8];       // coset: h^(+-M j1)
};
template <class FrP, int LG>
__global__ void __launch_bounds__(256) k_ntt_cross(Fp<FrP>* __restrict__ data, CrossArgs a) {
    using F = Fp<FrP>;
    constexpr int G = 1 << LG;
    const uint32_t B = 1u << a.logB;
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const uint64_t j2 = ((uint64_t)a.rank << a.logB) + c;
    const F* t_lo = reinterpret_cast<const F*>(a.t_lo);
    const F* t_hi = reinterpret_cast<const F*>(a.t_hi);
    const F* w_small = reinterpret_cast<const F*>(a.w_small);
    F v[G];
#pragma unroll
    for (int j = 0; j < G; j++) {
        v[j] = data[(size_t)j * B + c];
        if (a.to_mont) v[j] = zl::to_mont(v[j]);
    }
    // tw = cst * (coset ? h^(+-j2) : 1), step = w_N^(+-j2): row k1 is scaled by tw * step^k1
    F tw, step = twiddle2(t_lo, t_hi, a.L, j2);
#pragma unroll
    for (int w = 0; w < F::N; w++) tw.l[w] = a.cst[w];
    if (a.coset) tw = zl::mul(tw, twiddle2(reinterpret_cast<const F*>(a.g_lo), reinterpret_cast<const F*>(a.g_hi), a.L, j2));
    if (a.inverse) {
#pragma unroll
        for (int k = 0; k < G; k++) {
            v[k] = zl::mul(v[k], tw);
            tw = zl::mul(tw, step);
        }
    } else if (a.coset) {
#pragma unroll
        for (int j = 1; j < G; j++) {
            F r;
#pragma unroll
            for (int w = 0; w < F::N; w++) r.l[w] = a.row[j][w];
            v[j] = zl::mul(v[j], r);
        }
    }
    // G-point DIF butterflies; output k sits at position bitrev(k)
#pragma unroll
    for (int hl = LG - 1; hl >= 0; hl--) {
        const int h = 1 << hl;
#pragma unroll
        for (int q = 0; q < G / 2; q++) {
            const int k = q & (h - 1), i = ((q >> hl) << (hl + 1)) | k;
            const F u = v[i], x = v[i + h];
            v[i] = zl::add(u, x);
            F d = zl::sub(u, x);
            if (k != 0) d = zl::mul(d, w_small[k << (LG - 1 - hl)]);
            v[i + h] = d;
        }
    }
#pragma unroll
    for (int k = 0; k < G; k++) {
        int pos = 0;
#pragma unroll
        for (int b = 0; b < LG; b++) pos |= ((k >> b) & 1) << (LG - 1 - b);
        F x = v[pos];
        if (!a.inverse) {
            x = zl::mul(x, tw);
            tw = zl::mul(tw, step);
        } else if (a.coset && k != 0) {
            F r;
#pragma unroll
            for (int w = 0; w < F::N; w++) r.l[w] = a.row[k][w];
            x = zl::mul(x, r);
        }
        if (a.from_mont) x = zl::from_mont(x);
        data[(size_t)k * B + c] = x;
    }
}

template <class FrP>
static int ntt_cross_t(zl_ctx* ctx, int curve, void* d_data, unsigned n, unsigned lg, unsigned rank, unsigned flags) {
    using F = Fp<FrP>;
    if (n > (unsigned)FrP::TWO_ADICITY || n > 32 || lg < 1 || lg > 4 || 2 * lg > n || rank >= (1u << lg)) return ZL_EINVAL;
    const bool inverse = flags & ZL_INVERSE, coset = flags & ZL_COSET;
    const bool mont_in = flags & (ZL_MONT | ZL_MONT_IN), mont_out = flags & (ZL_MONT | ZL_MONT_OUT);
    zl_twiddles* tw;
    int rc;
    if ((rc = ntt_tables<FrP>(ctx, curve, n, inverse, &tw))) return rc;
    const unsigned L = tw->lo_bits;
    CrossArgs a{};
    a.n_log = n;
    a.lg = lg;
    a.logB = n - 2 * lg;
    a.L = L;
    a.rank = rank;
    a.inverse = inverse;
    a.coset = coset;
    a.to_mont = !mont_in;
    a.from_mont = !mont_out;
    const F* t_lo = reinterpret_cast<const F*>(tw->d_lo);
    const F* t_hi = reinterpret_cast<const F*>(tw->d_hi);
    a.t_lo = t_lo;
    a.t_hi = t_hi;
    a.g_lo = t_hi + ((size_t)1 << (n - L));
    a.g_hi = reinterpret_cast<const F*>(a.g_lo) + ((size_t)1 << L);
    a.w_small = reinterpret_cast<const F*>(tw->d_small) + ((size_t)1 << (lg - 1));
    F cst = F::one();
    if (inverse) cst = coset ? zl::from_u64<FrP>(1ull << (n - lg)) : zl::inv(zl::from_u64<FrP>(1ull << lg));
    for (int w = 0; w < F::N; w++) a.cst[w] = cst.l[w];
    if (coset) {
        F g;
        for (int i = 0; i < F::N; i++) g.l[i] = FrP::generator(i);
        if (inverse) g = zl::inv(g);
        F gM = g;
        for (unsigned i = 0; i < n - lg; i++) gM = zl::sqr(gM);  // h^(+-M)
        F acc = F::one();
        for (unsigned j = 0; j < (1u << lg); j++) {
            for (int w = 0; w < F::N; w++) a.row[j][w] = acc.l[w];
            acc = zl::mul(acc, gM);
        }
    }
    F* data = reinterpret_cast<F*>(d_data);
    const uint32_t B = 1u << a.logB;
    const dim3 grid((B + 255) / 256), block(256);
    hipStream_t st = ctx->stream;
    switch (lg) {
        case 1: hipLaunchKernelGGL((k_ntt_cross<FrP, 1>), grid, block, 0, st, data, a); break;
        case 2: hipLaunchKernelGGL((k_ntt_cross<FrP, 2>), grid, block, 0, st, data, a); break;
        case 3: hipLaunchKernelGGL((k_ntt_cross<FrP, 3>), grid, block, 0, st, data, a); break;
        default: hipLaunchKernelGGL((k_ntt_cross<FrP, 4>), grid, block, 0, st, data, a); break;
    }
    ZL_HIP(ctx, hipGetLastError());
    return ZL_OK;
}
int zl_ntt_cross_run(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned log_g, unsigned rank, unsigned flags) {
    if (curve == ZL_BLS12_381) return ntt_cross_t<BLS12_381_Fr>(ctx, curve, d_data, log_n, log_g, rank, flags);
    if (curve == ZL_BN254) return ntt_cross_t<BN254_Fr>(ctx, curve, d_data, log_n, log_g, rank, flags);
    return ZL_EINVAL;
}

int zl_ntt_run_batch(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned flags, unsigned count, size_t stride_bytes) {
    if (count == 0) return ZL_OK;
    if (count > 65535 || (count > 1 && stride_bytes < ((size_t)32 << log_n))) return ZL_EINVAL;
    if (curve == ZL_BLS12_381) return ntt_run_t<BLS12_381_Fr>(ctx, curve, d_data, log_n, flags, count, stride_bytes);
    if (curve == ZL_BN254) return ntt_run_t<BN254_Fr>(ctx, curve, d_data, log_n, flags, count, stride_bytes);
    return ZL_EINVAL;
}
int zl_ntt_run(zl_ctx* ctx, int curve, void* d_data, unsigned log_n, unsigned flags) {
    if (curve == ZL_BLS12_381) return ntt_run_t<BLS12_381_Fr>(ctx, curve, d_data, log_n, flags);
    if (curve == ZL_BN254) return ntt_run_t<BN254_Fr>(ctx, curve, d_data, log_n, flags);
    return ZL_EINVAL;
}
void zl_ntt_free(zl_ctx* ctx) {
    for (auto& kv : ctx->twiddles) {
        if (kv.second.d_lo) (void)hipFree(kv.second.d_lo);
        if (kv.second.d_last) (void)hipFree(kv.second.d_last);
        for (void* r : kv.second.d_row) if (r) (void)hipFree(r);
        if (kv.second.d_small_limbs) (void)hipFree(kv.second.d_small_limbs);
    }
    ctx->twiddles.clear();
    ctx->ntt_last_bytes = 0;
}
