// tools/ubench2.hip -- round 5: what the NON-mad instructions of the product scan cost on gfx950, at the occupancy of the accumulation kernel.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o /tmp/ubench2 ; run on the GPU box.
// Every wave times ITERS x 64 instructions with s_memtime (shader cycles) AND s_memrealtime (100 MHz): cycles per wave-instruction per SIMD
// = wave cycles / instructions / waves per SIMD, and the effective clock of the launch.  Grid = CUs x 4 x W blocks of one wave (as k_msm_accumulate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define DEF_KERNEL(name, N_PER_BODY, BODY)                                                         \
    __global__ void __launch_bounds__(64) name(uint64_t* out, uint32_t seed, int iters) {          \
        uint32_t a0 = seed * (threadIdx.x + 1) + blockIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3; \
        uint64_t d0 = ((uint64_t)a0 << 20) | a1, d1 = ((uint64_t)a1 << 21) | a2, d2 = ((uint64_t)a2 << 19) | a3, d3 = ((uint64_t)a3 << 22) | a0; \
        uint64_t t0 = __builtin_readcyclecounter(), w0 = __builtin_amdgcn_s_memrealtime();         \
        for (int i = 0; i < iters; i++) { REP16(BODY) }                                            \
        uint64_t t1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();         \
        uint64_t sink = d0 ^ d1 ^ d2 ^ d3 ^ a0 ^ a1 ^ a2 ^ a3;                                     \
        if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; } \
        if (sink == 0x1234567887654321ull) out[0] = sink;                                          \
    }                                                                                              \
    static const int name##_n = 16 * (N_PER_BODY);

// four independent chains of one instruction
DEF_KERNEL(k_mad, 4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
DEF_KERNEL(k_mad_dep, 4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %0, vcc, %5, %6, %0\n v_mad_u64_u32 %0, vcc, %6, %7, %0\n v_mad_u64_u32 %0, vcc, %7, %4, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
DEF_KERNEL(k_mad_sgpr, 4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, %0\n v_mad_u64_u32 %0, vcc, %5, %8, %0\n v_mad_u64_u32 %0, vcc, %6, %8, %0\n v_mad_u64_u32 %0, vcc, %7, %8, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(seed) : "vcc");)
DEF_KERNEL(k_shr64, 4, asm volatile("v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
DEF_KERNEL(k_alignbit, 4, asm volatile("v_alignbit_b32 %0, %1, %0, 28\n v_alignbit_b32 %1, %2, %1, 28\n v_alignbit_b32 %2, %3, %2, 28\n v_alignbit_b32 %3, %0, %3, 28" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_shr32, 4, asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_and, 4, asm volatile("v_and_b32 %0, 0xfffffff, %0\n v_and_b32 %1, 0xfffffff, %1\n v_and_b32 %2, 0xfffffff, %2\n v_and_b32 %3, 0xfffffff, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_add, 4, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_add3, 4, asm volatile("v_add3_u32 %0, %0, %1, %4\n v_add3_u32 %1, %1, %2, %4\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %0, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_mul_lo, 4, asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
DEF_KERNEL(k_mad_i64, 4, asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %5, %6, %1\n v_mad_i64_i32 %2, vcc, %6, %7, %2\n v_mad_i64_i32 %3, vcc, %7, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
DEF_KERNEL(k_mov, 4, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_bfe, 4, asm volatile("v_bfe_u32 %0, %0, 1, 28\n v_bfe_u32 %1, %1, 1, 28\n v_bfe_u32 %2, %2, 1, 28\n v_bfe_u32 %3, %3, 1, 28" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
DEF_KERNEL(k_and_or, 4, asm volatile("v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(seed));)
// the dependent chain with an s_nop after every 4 mads (hipcc pads every inline-asm statement of the product scans with one)
DEF_KERNEL(k_mad_dep_nop, 4, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %0, vcc, %5, %6, %0\n v_mad_u64_u32 %0, vcc, %6, %7, %0\n v_mad_u64_u32 %0, vcc, %7, %4, %0\n s_nop 0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
// one product column as gen_mul28.py emits it: 3 mads, m = lo * INV & mask, mad, 64-bit shift   (6 VALU instructions, 2 nops)
DEF_KERNEL(k_col_now, 7, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %0, vcc, %4, %2, %0\n s_nop 0\n v_mul_lo_u32 %1, %1, %2\n v_and_b32 %1, 0xfffffff, %1\n v_mad_u64_u32 %0, vcc, %1, %4, %0\n s_nop 0\n v_lshrrev_b64 %0, 28, %0" : "+v"(d0), "+v"(a1) : "v"(a0), "v"(a2), "v"(a3) : "vcc");)
// the same column with the shift as two 32-bit instructions (on other registers here: the issue cost is what is measured)
DEF_KERNEL(k_col_split, 8, asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %0, vcc, %5, %4, %0\n v_mad_u64_u32 %0, vcc, %4, %4, %0\n s_nop 0\n v_mul_lo_u32 %1, %1, %4\n v_and_b32 %1, 0xfffffff, %1\n v_mad_u64_u32 %0, vcc, %1, %5, %0\n s_nop 0\n v_alignbit_b32 %2, %3, %2, 28\n v_lshrrev_b32 %3, 28, %3" : "+v"(d0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a0), "v"(seed) : "vcc");)
// ... and without the two s_nop
DEF_KERNEL(k_col_nonop, 7, asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %0, vcc, %3, %4, %0\n v_mad_u64_u32 %0, vcc, %4, %2, %0\n v_mul_lo_u32 %1, %1, %2\n v_and_b32 %1, 0xfffffff, %1\n v_mad_u64_u32 %0, vcc, %1, %4, %0\n v_lshrrev_b64 %0, 28, %0" : "+v"(d0), "+v"(a1) : "v"(a0), "v"(a2), "v"(a3) : "vcc");)

typedef void (*kern_t)(uint64_t*, uint32_t, int);
struct Entry { const char* name; kern_t k; int n; };
#define E(label, k) {label, k, k##_n}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    Entry es[] = {E("v_mad_u64_u32 (4 chains)", k_mad), E("v_mad_u64_u32 (1 chain)", k_mad_dep), E("v_mad_u64_u32 v,s (1 chain)", k_mad_sgpr), E("v_mad_i64_i32", k_mad_i64), E("v_mul_lo_u32", k_mul_lo),
                  E("v_lshrrev_b64", k_shr64), E("v_alignbit_b32", k_alignbit), E("v_lshrrev_b32", k_shr32), E("v_and_b32 literal", k_and), E("v_bfe_u32", k_bfe), E("v_and_or_b32", k_and_or),
                  E("v_add_u32", k_add), E("v_add3_u32", k_add3), E("v_mov_b32", k_mov), E("4 dep mads + s_nop 0 (per mad)", k_mad_dep_nop),
                  E("column: 4 mad, mul_lo, and, shr64 (7 instr)", k_col_now), E("column: ... alignbit + shr32 (8 instr)", k_col_split), E("column: as the first, no s_nop (7 instr)", k_col_nonop)};
    const int iters = 4000;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * (1 << 16)));
    printf("device CUs=%d.  Columns: waves per SIMD 1, 2, 3, 4 -> SIMD cycles per wave-instruction (s_memtime) [effective GHz]\n", prop.multiProcessorCount);
    for (auto& e : es) {
        printf("%-48s", e.name);
        for (int wps : {1, 2, 3, 4}) {
            const int blocks = prop.multiProcessorCount * 4 * wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, 200);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, iters);
            CHECK(hipDeviceSynchronize());
            std::vector<uint64_t> h(2 * blocks);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(uint64_t) * 2 * blocks, hipMemcpyDeviceToHost));
            double sc = 0, sw = 0;
            for (int i = 0; i < blocks; i++) { sc += (double)h[2 * i]; sw += (double)h[2 * i + 1]; }
            printf("  %6.2f [%4.2f]", sc / blocks / ((double)iters * e.n) / wps, sc / sw * 0.1);
        }
        printf("\n");
    }
    return 0;
}
