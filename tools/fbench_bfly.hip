// butterfly throughput of the current 8x32-bit Fr arithmetic (mul + add + sub per step), for comparison with fbench28r
#include "../openzl_amd/csrc/zl_field.h"
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using Fr = Fp<BLS12_381_Fr>;
__global__ void k_chain(Fr* a, const Fr* b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr x = a[i], y = b[i], w = b[i ^ 1];
    for (int k = 0; k < iters; k++) { Fr t = zl::mul(y, w); Fr u = zl::add(x, t); y = zl::sub(x, t); x = u; }
    a[i] = zl::add(x, y);
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int wps : {2, 4, 8}) {
        int threads = 256, blocks = prop.multiProcessorCount * wps;
        size_t n = (size_t)threads * blocks;
        Fr *a, *b; CHECK(hipMalloc(&a, n * sizeof(Fr))); CHECK(hipMalloc(&b, n * sizeof(Fr)));
        CHECK(hipMemset(a, 0x05, n * sizeof(Fr))); CHECK(hipMemset(b, 0x03, n * sizeof(Fr)));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, 4);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int iters = 4000;
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain, dim3(blocks), dim3(threads), 0, 0, a, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("Fr 8x32 butterfly(mul+add+sub, inlined) waves/SIMD=%d  %8.3f ms  %9.2f G/s\n", wps, ms, (double)n * iters / (ms * 1e-3) / 1e9);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
