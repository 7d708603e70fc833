// tools/fbench28_pair.hip -- experiment: does the s_nop that hipcc places after every inline-asm statement whose result is read by the next
// instruction (3 per column of the single-chain scan) cost throughput?  Two independent products per iteration, either as two single-chain
// scans one after the other (MODE 0) or interleaved statement by statement (MODE 1: tools/gen_mul28x2.py, no result is
// read by its successor).  Result (profiles/r02_fbench28_pair.log): +12 % at one wave per SIMD, +9 % at two, nothing from four waves on.
#include "../openzl_amd/csrc/zl_field28.h"
#include "mul28x2_asm.h"
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using A = BLS12_381_Fq28;
using F = Fp28<A, BLS12_381_Fq>;
template <int MODE>
__global__ void __launch_bounds__(64) k_chain(F* a, const F* b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x1 = a[i], x2 = b[i], y = b[i ^ 1];
    for (int k = 0; k < iters; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
        F r1 = x1, r2 = x2;
        if (MODE == 0) { mul28_asm<A>(r1.l, x1.l, y.l); mul28_asm<A>(r2.l, x2.l, y.l); }
        else mul28x2_asm<A>(r1.l, x1.l, y.l, r2.l, x2.l, y.l);
        x1 = r1; x2 = r2;
#endif
    }
    a[i] = x1;
    a[i].l[0] ^= x2.l[0]; a[i].l[13] ^= x2.l[13];
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int mode : {0, 1}) for (int wps : {1, 2, 3, 4, 5}) {
        int threads = 64, blocks = prop.multiProcessorCount * 4 * wps;
        size_t n = (size_t)threads * blocks;
        F *a, *b; CHECK(hipMalloc(&a, n * sizeof(F))); CHECK(hipMalloc(&b, n * sizeof(F)));
        CHECK(hipMemset(a, 0x05, n * sizeof(F))); CHECK(hipMemset(b, 0x03, n * sizeof(F)));
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, a, b, 4); else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, a, b, 4);
        CHECK(hipDeviceSynchronize());
        uint32_t h[16]; CHECK(hipMemcpy(h, a, 64, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int iters = 2000;
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, a, b, iters); else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, a, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s waves/SIMD=%d  %8.3f ms  %8.2f G mul/s   (after 4 iterations: l[0]=%07x l[13]=%07x)\n", mode ? "interleaved pair     " : "two single chains    ", wps, ms,
               2.0 * (double)n * iters / (ms * 1e-3) / 1e9, h[0], h[13]);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
