// tools/fbench.hip -- field-multiplication throughput of zl_field.h on the GPU (chip-wide Gmul/s).
#include "../openzl_amd/csrc/zl_field.h"
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class F, int OP>
__global__ void k_chain(F* a, const F* b, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    F x = a[i], y = b[i];
    for (int k = 0; k < iters; k++) {
        if (OP == 0) x = zl::mul(x, y);
        if (OP == 1) { x = zl::add(x, y); y = zl::sub(y, x); }
        if (OP == 2) x = zl::sqr(x);
    }
    a[i] = x;
}
template <class F, int OP>
int run(const char* name, int wps_list_n, int iters, int opsper) {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    for (int wps : {1, 2, 4, 8}) {
        int threads = 256, blocks = prop.multiProcessorCount * wps;
        size_t n = (size_t)threads * blocks;
        F *a, *b; CHECK(hipMalloc(&a, n * sizeof(F))); CHECK(hipMalloc(&b, n * sizeof(F)));
        CHECK(hipMemset(a, 0x11, n * sizeof(F))); CHECK(hipMemset(b, 0x07, n * sizeof(F)));
        hipLaunchKernelGGL((k_chain<F, OP>), dim3(blocks), dim3(threads), 0, 0, a, b, 4);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain<F, OP>), dim3(blocks), dim3(threads), 0, 0, a, b, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s waves/SIMD=%d  %8.3f ms  %9.2f Gop/s\n", name, wps, ms, (double)n * iters * opsper / (ms * 1e-3) / 1e9);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
int main() {
    using Fq = Fp<BLS12_381_Fq>; using Fr = Fp<BLS12_381_Fr>;
    run<Fq, 0>("bls12_381 Fq mul (12x32)", 0, 2000, 1);
    run<Fr, 0>("bls12_381 Fr mul (8x32)", 0, 4000, 1);
    run<Fq, 1>("bls12_381 Fq add+sub", 0, 8000, 2);
    return 0;
}
