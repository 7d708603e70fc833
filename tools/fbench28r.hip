// tools/fbench28r.hip -- experiment: BLS12-381 Fr on 10 unsaturated 28-bit limbs (R = 2^280) vs the 8x32-bit multiplier.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int L = 10;
constexpr uint32_t MASK = (1u << 28) - 1;
struct F28 { uint32_t l[L]; };
__constant__ uint32_t c_mod[L];
__constant__ uint32_t c_inv;
__device__ __forceinline__ F28 mul28(const F28& a, const F28& b) {
    uint32_t m[L];
    F28 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * c_mod[k - i];
        m[k] = ((uint32_t)acc * c_inv) & MASK;
        acc += (uint64_t)m[k] * c_mod[0];
        acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * c_mod[k - i];
        r.l[k - L] = (uint32_t)acc & MASK;
        acc >>= 28;
    }
    return r;
}
// butterfly-like mix: one mul + lazy add + lazy sub (bias) per step
__global__ void k_chain(F28* a, const F28* b, int iters, int mode) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F28 x = a[i], y = b[i], w = b[i ^ 1];
    for (int k = 0; k < iters; k++) {
        if (mode == 0) x = mul28(x, y);
        else {
            F28 t = mul28(y, w);
            F28 u, v;
#pragma unroll
            for (int j = 0; j < L; j++) { u.l[j] = x.l[j] + t.l[j]; v.l[j] = x.l[j] + (0x10000000u) - t.l[j]; }
#pragma unroll
            for (int j = 0; j < L - 1; j++) { u.l[j + 1] += u.l[j] >> 28; u.l[j] &= MASK; v.l[j + 1] += v.l[j] >> 28; v.l[j] &= MASK; }
            x = u; y = v;
        }
    }
    a[i] = x;
    if (mode) a[i].l[0] ^= y.l[0];
}
int main() {
    const char* qhex = "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001";
    uint32_t mod[L] = {0};
    int nh = 64;
    for (int bit = 0; bit < 256; bit++) {
        char ch = qhex[nh - 1 - bit / 4];
        int v = (ch >= 'a') ? ch - 'a' + 10 : ch - '0';
        if ((v >> (bit % 4)) & 1) mod[bit / 28] |= 1u << (bit % 28);
    }
    uint32_t inv = 1;
    for (int i = 0; i < 5; i++) inv *= 2 - mod[0] * inv;
    inv = (0u - inv) & MASK;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_mod), mod, sizeof mod));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_inv), &inv, sizeof inv));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int mode : {0, 1}) for (int wps : {2, 4, 8}) {
        int threads = 256, blocks = prop.multiProcessorCount * wps;
        size_t n = (size_t)threads * blocks;
        F28 *a, *b; CHECK(hipMalloc(&a, n * sizeof(F28))); CHECK(hipMalloc(&b, n * sizeof(F28)));
        CHECK(hipMemset(a, 0x05, n * sizeof(F28))); CHECK(hipMemset(b, 0x03, n * sizeof(F28)));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, 4, mode);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int iters = 4000;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, iters, mode);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("Fr 10x28 %s waves/SIMD=%d  %8.3f ms  %9.2f G/s\n", mode ? "butterfly(mul+add+sub)" : "mul                   ", wps, ms, (double)n * iters / (ms * 1e-3) / 1e9);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
