// tools/fbench28_asm.hip -- experiment: compiler-scheduled vs single-chain inline-asm 14 x 28-bit Montgomery product scan (same harness as
// fbench28.hip, constants from zl_params.h so that the scan can use literal / SGPR modulus limbs).
#include "../openzl_amd/csrc/zl_field28.h"
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using A = BLS12_381_Fq28;
using F = Fp28<A, BLS12_381_Fq>;
template <int MODE>
__global__ void k_chain(F* a, const F* b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x = a[i], y = b[i];
    for (int k = 0; k < iters; k++) {
#if defined(__HIP_DEVICE_COMPILE__)  // the asm scan (openzl_amd/csrc/zl_mul28_gfx950.h, via zl_field28.h) exists in the device pass only
        if (MODE == 0) x = zl::mul_body28(x, y);
        else { F r = x; mul28_asm<A>(r.l, x.l, y.l); x = r; }
#endif
    }
    a[i] = x;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int mode : {0, 1}) for (int wps : {1, 2, 4}) {
        int threads = 64, blocks = prop.multiProcessorCount * 4 * wps;
        size_t n = (size_t)threads * blocks;
        F *a, *b; CHECK(hipMalloc(&a, n * sizeof(F))); CHECK(hipMalloc(&b, n * sizeof(F)));
        CHECK(hipMemset(a, 0x05, n * sizeof(F))); CHECK(hipMemset(b, 0x03, n * sizeof(F)));
        // same inputs for both modes: check that they agree after a few iterations
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, a, b, 4); else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, a, b, 4);
        CHECK(hipDeviceSynchronize());
        uint32_t h[16]; CHECK(hipMemcpy(h, a, 64, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int iters = 4000;
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(threads), 0, 0, a, b, iters); else hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(threads), 0, 0, a, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s waves/SIMD=%d  %8.3f ms  %8.2f G mul/s   (after 4 muls: l[0]=%07x l[13]=%07x)\n", mode ? "asm single chain " : "compiler (C++)   ", wps, ms,
               (double)n * iters / (ms * 1e-3) / 1e9, h[0], h[13]);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
