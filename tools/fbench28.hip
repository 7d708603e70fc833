// tools/fbench28.hip -- experiment: unsaturated 28-bit-limb Montgomery multiplier (14 limbs, R = 2^392) vs the 32-bit-limb one.
// Column sums of <= 28 products < 2^56 never overflow a 64-bit accumulator, so the inner loop is pure v_mad_u64_u32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int L = 14;
constexpr uint32_t MASK = (1u << 28) - 1;
struct F28 { uint32_t l[L]; };
// BLS12-381 q in 28-bit limbs and -q^-1 mod 2^28 are filled on the host
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
    return r;  // < 2q, limbs normalised (lazy: no final subtraction, 4q < 2^392)
}
__global__ void k_chain(F28* a, const F28* b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F28 x = a[i], y = b[i];
    for (int k = 0; k < iters; k++) x = mul28(x, y);
    a[i] = x;
}
int main() {
    // q = BLS12-381 base field modulus
    const char* qhex = "1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab";
    unsigned __int128 dummy = 0; (void)dummy;
    // parse hex into 28-bit limbs
    uint32_t mod[L] = {0};
    int nh = 96;
    for (int bit = 0; bit < 384; bit++) {
        int hexpos = nh - 1 - bit / 4;
        char ch = qhex[hexpos];
        int v = (ch >= 'a') ? ch - 'a' + 10 : ch - '0';
        if ((v >> (bit % 4)) & 1) mod[bit / 28] |= 1u << (bit % 28);
    }
    uint32_t inv = 1;  // -q^-1 mod 2^28 by Newton
    for (int i = 0; i < 5; i++) inv *= 2 - mod[0] * inv;
    inv = (0u - inv) & MASK;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_mod), mod, sizeof mod));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_inv), &inv, sizeof inv));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int wps : {1, 2, 4, 8}) {
        int threads = 256, blocks = prop.multiProcessorCount * wps;
        size_t n = (size_t)threads * blocks;
        F28 *a, *b; CHECK(hipMalloc(&a, n * sizeof(F28))); CHECK(hipMalloc(&b, n * sizeof(F28)));
        CHECK(hipMemset(a, 0x05, n * sizeof(F28))); CHECK(hipMemset(b, 0x03, n * sizeof(F28)));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, 4);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int iters = 2000;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("Fq mul 14x28-bit lazy     waves/SIMD=%d  %8.3f ms  %9.2f Gop/s\n", wps, ms, (double)n * iters / (ms * 1e-3) / 1e9);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
