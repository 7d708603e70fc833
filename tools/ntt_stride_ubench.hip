// ntt_stride_ubench.hip -- the memory side of a TWO-pass 2^24 NTT (2^12 x 2^12), measured.
// In a two-pass factorisation the first pass transforms 4096-point columns whose elements lie 2^12 elements = 128 KiB apart.  A tile of C
// columns is 4096 x C x 32 B = C x 128 KiB of LDS, so C = 1 (160 KiB per CU; the 2^11 butterfly roots, 64 KiB, no longer fit beside it): every
// global read is ONE 32-byte element at a 128-KiB stride, one workgroup per CU.  This benchmark moves 2^24 elements with exactly that pattern
// (strided 32-B gather -> LDS -> the same strided scatter, no arithmetic) and, for comparison, with the 8-column rows of the three-pass
// kernel (256-B segments, 64-KiB tiles, two workgroups per CU).  hipcc --offload-arch=gfx950 -O3 tools/ntt_stride_ubench.hip -o tools/ntt_stride_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
struct alignas(16) El { unsigned v[8]; };
// tile = R rows x C columns; element (r, c) of tile t lives at base(t) + r * row_stride + c
template <int LOGR, int LOGC>
__global__ void __launch_bounds__(512) k_move(const El* __restrict__ in, El* __restrict__ out, unsigned n_log) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    El* sh = reinterpret_cast<El*>(smem);
    constexpr unsigned R = 1u << LOGR, C = 1u << LOGC;
    const unsigned stride_log = n_log - LOGR;                    // elements between consecutive rows
    const unsigned long long lo_blocks = (1ull << stride_log) >> LOGC;
    const unsigned long long tile = blockIdx.x, lo0 = (tile % lo_blocks) << LOGC;
    const unsigned long long base = lo0;                         // one factor only: hi = 0
    for (unsigned idx = threadIdx.x; idx < R * C; idx += 512) {
        const unsigned c = idx & (C - 1), r = idx >> LOGC;
        sh[idx] = in[base + ((unsigned long long)r << stride_log) + c];
    }
    __syncthreads();
    for (unsigned idx = threadIdx.x; idx < R * C; idx += 512) {
        const unsigned c = idx & (C - 1), r = idx >> LOGC;
        El e = sh[(idx * 33u) & (R * C - 1)];                    // any permutation inside the tile: the butterflies' stand-in
        e.v[0] ^= r;
        out[base + ((unsigned long long)r << stride_log) + c] = e;
    }
}
template <int LOGR, int LOGC>
static float run(const El* in, El* out, unsigned n_log, const char* what) {
    const size_t lds = sizeof(El) << (LOGR + LOGC);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_move<LOGR, LOGC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const unsigned tiles = 1u << (n_log - LOGR - LOGC);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_move<LOGR, LOGC>), dim3(tiles), dim3(512), lds, 0, in, out, n_log);
    hipEventRecord(a, 0);
    const int reps = 10;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_move<LOGR, LOGC>), dim3(tiles), dim3(512), lds, 0, in, out, n_log);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", what); return 0; }
    printf("%-58s %7.3f ms  %7.1f GB/s (32 B read + 32 B written per element)\n", what, ms, 64.0 * (1u << n_log) / ms / 1e6);
    return ms;
}
int main() {
    const unsigned n_log = 24;
    El *in, *out;
    hipMalloc(&in, sizeof(El) << n_log);
    hipMalloc(&out, sizeof(El) << n_log);
    hipMemset(in, 1, sizeof(El) << n_log);
    run<8, 3>(in, out, n_log, "three-pass shape: 256 rows x 8 columns (256-B rows, 64 KiB)");
    run<12, 0>(in, out, n_log, "two-pass shape: 4096 rows x 1 column (32-B rows, 128 KiB)");
    run<11, 1>(in, out, n_log, "2048 rows x 2 columns (64-B rows, 128 KiB)");
    run<10, 2>(in, out, n_log, "1024 rows x 4 columns (128-B rows, 128 KiB)");
    return 0;
}
