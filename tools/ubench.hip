// tools/ubench.hip -- gfx950 VALU issue-rate microbenchmark for the instructions the field arithmetic uses.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
// Prints cycles per wave-instruction (s_memtime ticks = shader cycles) at 1, 2, 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// each kernel: ITERS x 64 instructions on 4 independent dependency chains
#define DEF_KERNEL(name, ASM4)                                                                     \
    __global__ void name(uint64_t* out, uint32_t seed, int iters) {                               \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3;      \
        uint64_t d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                              \
        double f0 = a0 * 1e-3, f1 = a1 * 1e-3, f2 = a2 * 1e-3, f3 = a3 * 1e-3;                    \
        uint64_t t0 = __builtin_readcyclecounter();                                                \
        for (int i = 0; i < iters; i++) { REP16(ASM4) }                                            \
        uint64_t t1 = __builtin_readcyclecounter();                                                \
        uint64_t sink = d0 ^ d1 ^ d2 ^ d3 ^ a0 ^ a1 ^ a2 ^ a3 ^ (uint64_t)(f0 + f1 + f2 + f3);    \
        if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;   \
        if (sink == 0x1234567887654321ull) out[0] = sink;                                          \
    }

DEF_KERNEL(k_mad_u64_u32,
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
DEF_KERNEL(k_mul_lo_u32,
    asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_mul_hi_u32,
    asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_mad_u32_u24,
    asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_mul_hi_u32_u24,
    asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_add_u32,
    asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_add_co_addc,
    asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed) : "vcc");)
DEF_KERNEL(k_lshl_add_u64,
    asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
DEF_KERNEL(k_mov_b32,
    asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_fma_f64,
    asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
DEF_KERNEL(k_add_f64,
    asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
DEF_KERNEL(k_mul_f64,
    asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));)
DEF_KERNEL(k_fma_f32,
    asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_mad_u64_dep,
    asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %0, vcc, %5, %6, %0\n v_mad_u64_u32 %0, vcc, %6, %7, %0\n v_mad_u64_u32 %0, vcc, %7, %4, %0"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
DEF_KERNEL(k_cvt_f64_u32,
    asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %5\n v_cvt_f64_u32 %2, %6\n v_cvt_f64_u32 %3, %7"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
DEF_KERNEL(k_mad_i32_i24,
    asm volatile("v_mad_i32_i24 %0, %0, %4, %1\n v_mad_i32_i24 %1, %1, %4, %2\n v_mad_i32_i24 %2, %2, %4, %3\n v_mad_i32_i24 %3, %3, %4, %0"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)

typedef void (*kern_t)(uint64_t*, uint32_t, int);
struct Entry { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs=%d, clock=%d kHz, LDS/block=%zu, regs/block=%d, L2=%d\n", prop.name, prop.multiProcessorCount,
           prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock, prop.l2CacheSize);
    Entry es[] = {{"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(dep)", k_mad_u64_dep}, {"v_mul_lo_u32", k_mul_lo_u32},
                  {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
                  {"v_mad_i32_i24", k_mad_i32_i24}, {"v_add_u32", k_add_u32}, {"v_add_co+addc", k_add_co_addc},
                  {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mov_b32", k_mov_b32}, {"v_fma_f32", k_fma_f32},
                  {"v_fma_f64", k_fma_f64}, {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_cvt_f64_u32", k_cvt_f64_u32}};
    const int iters = 2000;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * 1 << 20));
    printf("%-22s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD; wall-derived Ginstr/s chip-wide at 8 w/SIMD)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
    for (auto& e : es) {
        printf("%-22s", e.name);
        double ginstr = 0;
        for (int wps : {1, 2, 4, 8}) {
            int threads = 256 * wps;  // one block per CU: 4 SIMDs x wps waves
            int blocks = prop.multiProcessorCount;
            if (threads > 1024) { blocks *= threads / 1024; threads = 1024; }
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, 10);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, d_out, 12345u, iters);
            hipEventRecord(e1);
            CHECK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int nw = blocks * threads / 64;
            std::vector<uint64_t> h(nw);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(uint64_t) * nw, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : h) avg += (double)v; avg /= nw;
            double cyc_per_instr_per_simd = avg / (iters * 64.0) / wps;   // wave-cycles per instr, divided by waves sharing the SIMD
            printf(" %10.2f", cyc_per_instr_per_simd);
            ginstr = (double)nw * iters * 64.0 / (ms * 1e-3) / 1e9;
        }
        printf("   %8.1f Gwave-instr/s\n", ginstr);
    }
    return 0;
}
