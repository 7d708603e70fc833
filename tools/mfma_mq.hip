// mfma_mq.hip -- microbenchmark for north_star's "MFMA only if a packed-limb contraction actually wins, evidenced by rocprof":
// the m*q half of the 14 x 28-bit Montgomery product (the only half with a shared operand: q is the same for every lane) as an int8
// Toeplitz contraction on the matrix cores, INCLUDING the digit split of m and the fold-back of the i32 columns into 28-bit limbs,
// against the same product as 196 v_mad_u64_u32 (what zl_mul28_gfx950.h spends on it).
//   m = 56 digits of 7 bits, q = 55 digits of 7 bits; columns c_r = sum_{i+j=r} q_i m_j, r < 111:
//   C (128 x 64) = T (128 x 64, Toeplitz of q's digits) * M (64 x 64: one column of digits per lane)
//   = 4 M-tiles x 2 N-tiles x 2 K-steps = 16 v_mfma_i32_32x32x32_i8 per wave-level product.
// Both kernels produce the 28 product limbs of m*q per lane (checked against 128-bit host arithmetic), ITER dependent rounds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_mq.hip -o tools/mfma_mq && ./tools/mfma_mq
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const uint32_t Q28[14] = {0xfffaaabu, 0xfefffffu, 0x3ffffb9u, 0xfffeb15u, 0x6241eabu, 0xa0f6b0fu, 0xf6730d2u, 0xf38512bu,
                                 0x4774b84u, 0x4bacd76u, 0xba7b643u, 0xe69a4b1u, 0x1ea397fu, 0x001a011u};
__constant__ uint32_t dQ28[14];
#define M28 0xFFFFFFFu

// ---- reference path: 196 mads, product scanning with a 64-bit column accumulator (limbs < 2^28: no carries inside a column)
__device__ __forceinline__ void mq_mads(const uint32_t m[14], uint32_t out[28]) {
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 27; k++) {
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const int j = k - i;
            if (j >= 0 && j < 14) acc += (uint64_t)m[i] * dQ28[j];
        }
        out[k] = (uint32_t)acc & M28;
        acc >>= 28;
    }
    out[27] = (uint32_t)acc;
}
template <int ITER>
__global__ void __launch_bounds__(64) k_mads(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t m[14], r[28];
    for (int i = 0; i < 14; i++) m[i] = in[(size_t)t * 14 + i];
    for (int it = 0; it < ITER; it++) {
        mq_mads(m, r);
        if (it + 1 < ITER)
            for (int i = 0; i < 14; i++) m[i] = (r[i] ^ r[14 + i]) & M28;  // dependent rounds
    }
    for (int i = 0; i < 28; i++) out[(size_t)t * 28 + i] = r[i];
}

// ---- MFMA path ------------------------------------------------------------------------------------------------------------------
// slot (h, j) of a K-step s carries k = 32 s + 16 h + j for BOTH operands (h = lane >> 5, j = byte 0..15 of the 4 operand VGPRs): the
// hardware pairs equal slots, so any consistent assignment gives the same dot products.
__device__ __forceinline__ uint32_t digit7(const uint32_t m[14], int d) {  // d-th 7-bit digit of a 14 x 28-bit number (4 digits per limb)
    return (m[d >> 2] >> (7 * (d & 3))) & 0x7Fu;
}
__device__ __forceinline__ uint32_t swap32(uint32_t v) {  // value of the lane 32 away
    return (uint32_t)__shfl_xor((int)v, 32);
}
template <int ITER>
__global__ void __launch_bounds__(64) k_mfma(const uint32_t* __restrict__ in, const int8_t* __restrict__ toeplitz /* [128][64] */, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, h = lane >> 5, n = lane & 31;
    uint32_t m[14], r[28];
    for (int i = 0; i < 14; i++) m[i] = in[(size_t)t * 14 + i];
    // A fragments (constant): M-tile mt, K-step s: row = 32 mt + n, k = 32 s + 16 h + j
    v4i A[4][2];
    for (int mt = 0; mt < 4; mt++)
        for (int s = 0; s < 2; s++) {
            const int8_t* row = toeplitz + (size_t)(32 * mt + n) * 64 + 32 * s + 16 * h;
            A[mt][s] = *reinterpret_cast<const v4i*>(row);
        }
    for (int it = 0; it < ITER; it++) {
        // 1. digit split: own digits packed 4 per word, k = 0..55 (words 14, 15 = 0)
        uint32_t dw[16];
#pragma unroll
        for (int w = 0; w < 14; w++) dw[w] = digit7(m, 4 * w) | (digit7(m, 4 * w + 1) << 8) | (digit7(m, 4 * w + 2) << 16) | (digit7(m, 4 * w + 3) << 24);
        dw[14] = dw[15] = 0;
        // 2. B fragments.  N-tile 0 = columns of lanes 0..31, N-tile 1 = columns of lanes 32..63.  Lane (h, n), K-step s, needs the words
        //    [8 s + 4 h, +4) of column n (tile 0) / column 32 + n (tile 1): its own words for tile h, the partner lane's for tile 1 - h.
        v4i B[2][2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            uint32_t own[4], oth[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // the partner needs MY words at ITS h: [8 s + 4 (1 - h), +4); I need the partner's words at MY h
                const uint32_t mine_lo = dw[8 * s + q], mine_hi = dw[8 * s + 4 + q];
                own[q] = h ? mine_hi : mine_lo;
                oth[q] = swap32(h ? mine_lo : mine_hi);
            }
            // tile index == h -> own column; tile index != h -> partner's column
            B[0][s] = h == 0 ? v4i{(int)own[0], (int)own[1], (int)own[2], (int)own[3]} : v4i{(int)oth[0], (int)oth[1], (int)oth[2], (int)oth[3]};
            B[1][s] = h == 1 ? v4i{(int)own[0], (int)own[1], (int)own[2], (int)own[3]} : v4i{(int)oth[0], (int)oth[1], (int)oth[2], (int)oth[3]};
        }
        // 3. 16 MFMAs
        v16i C[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                c = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[mt][0], B[nt][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[mt][1], B[nt][1], c, 0, 0, 0);
                C[mt][nt] = c;
            }
        // 4. gather all 128 rows of MY column: C[mt][nt][reg] = row 32 mt + (reg & 3) + 8 (reg >> 2) + 4 h of column (32 nt + n).
        //    My column lives in tile nt = h; the other half of its rows is held by the partner lane (same tile), and I hold half of the
        //    partner's column in tile 1 - h: one exchange per register.
        uint32_t col[128];
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const uint32_t mine = (uint32_t)(h ? C[mt][1][reg] : C[mt][0][reg]);     // my column, my rows
                const uint32_t give = (uint32_t)(h ? C[mt][0][reg] : C[mt][1][reg]);     // partner's column, my rows
                const uint32_t got = swap32(give);                                        // my column, partner's rows
                const int base = 32 * mt + (reg & 3) + 8 * (reg >> 2);
                // my rows are base + 4 h, the partner's base + 4 (1 - h): select without dynamic indexing
                col[base + 0] = h ? got : mine;
                col[base + 4] = h ? mine : got;
            }
        // 5. fold: value = sum col[r] 2^(7 r); limb j collects rows 4 j .. 4 j + 3 (28 limbs; rows >= 111 are zero)
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 28; j++) {
            uint64_t v = carry + (uint64_t)col[4 * j] + ((uint64_t)col[4 * j + 1] << 7) + ((uint64_t)col[4 * j + 2] << 14) + ((uint64_t)col[4 * j + 3] << 21);
            r[j] = (uint32_t)v & M28;
            carry = v >> 28;
        }
        if (it + 1 < ITER)
            for (int i = 0; i < 14; i++) m[i] = (r[i] ^ r[14 + i]) & M28;
    }
    for (int i = 0; i < 28; i++) out[(size_t)t * 28 + i] = r[i];
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
static void host_rounds(uint32_t m[14], int iters, uint32_t r[28]) {
    for (int it = 0; it < iters; it++) {
        unsigned __int128 acc = 0;
        for (int k = 0; k < 28; k++) {
            for (int i = 0; i < 14; i++) {
                const int j = k - i;
                if (j >= 0 && j < 14) acc += (unsigned __int128)m[i] * Q28[j];
            }
            r[k] = (uint32_t)acc & M28;
            acc >>= 28;
        }
        if (it + 1 < iters)
            for (int i = 0; i < 14; i++) m[i] = (r[i] ^ r[14 + i]) & M28;
    }
}
template <class K, class... Args>
static float time_ms(K kern, dim3 grid, int reps, Args... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, args...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char** argv) {
    constexpr int ITER = 64;
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const size_t lanes = (size_t)256 * 4 * waves_per_simd * 64 * 8;  // 8 rounds of resident waves
    hipMemcpyToSymbol(HIP_SYMBOL(dQ28), Q28, sizeof Q28);
    std::vector<uint32_t> hin(lanes * 14);
    uint64_t x = 88172645463325252ull;
    for (auto& v : hin) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t)x & M28; }
    // Toeplitz of q's 7-bit digits: T[r][k] = qd[r - k]
    uint32_t qd[56] = {0};
    for (int d = 0; d < 55; d++) qd[d] = (Q28[d >> 2] >> (7 * (d & 3))) & 0x7F;
    std::vector<int8_t> T(128 * 64, 0);
    for (int r = 0; r < 128; r++)
        for (int k = 0; k < 56; k++)
            if (r - k >= 0 && r - k < 56) T[r * 64 + k] = (int8_t)qd[r - k];
    uint32_t *din, *dout1, *dout2;
    int8_t* dT;
    hipMalloc(&din, hin.size() * 4);
    hipMalloc(&dout1, lanes * 28 * 4);
    hipMalloc(&dout2, lanes * 28 * 4);
    hipMalloc(&dT, T.size());
    hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dT, T.data(), T.size(), hipMemcpyHostToDevice);
    const dim3 grid((unsigned)(lanes / 64));
    const float t_mads = time_ms(k_mads<ITER>, grid, 5, (const uint32_t*)din, dout1);
    const float t_mfma = time_ms(k_mfma<ITER>, grid, 5, (const uint32_t*)din, (const int8_t*)dT, dout2);
    std::vector<uint32_t> o1(lanes * 28), o2(lanes * 28);
    hipMemcpy(o1.data(), dout1, o1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o2.data(), dout2, o2.size() * 4, hipMemcpyDeviceToHost);
    size_t bad1 = 0, bad2 = 0;
    for (size_t t = 0; t < lanes; t += 997) {
        uint32_t m[14], r[28];
        for (int i = 0; i < 14; i++) m[i] = hin[t * 14 + i];
        host_rounds(m, ITER, r);
        for (int i = 0; i < 28; i++) { bad1 += o1[t * 28 + i] != r[i]; bad2 += o2[t * 28 + i] != r[i]; }
    }
    const double prods = (double)lanes * ITER;
    printf("m*q half of the 14x28-bit Montgomery product, %zu lanes x %d dependent rounds, launch sized for %d waves/SIMD resident\n", lanes, ITER, waves_per_simd);
    printf("  196 x v_mad_u64_u32            : %8.3f ms  %7.2f G products/s   mismatches vs host: %zu\n", t_mads, prods / t_mads / 1e6, bad1);
    printf("  int8 MFMA Toeplitz (16 mfma)   : %8.3f ms  %7.2f G products/s   mismatches vs host: %zu\n", t_mfma, prods / t_mfma / 1e6, bad2);
    printf("  ratio mfma / mads time: %.2f (> 1: the matrix-core formulation is slower, digit split + lane exchange + fold-back included)\n", t_mfma / t_mads);
    return (bad1 || bad2) ? 1 : 0;
}
