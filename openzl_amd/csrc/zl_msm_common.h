// zl_msm_common.h -- tunables shared by the units of the MSM (zl_msm_sort.hip, zl_msm_acc.hip, zl_msm_tail.hip, zl_msm.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// The per-group units are compiled once per group: -DZL_G=BlsG1|BnG1|BlsG2|BnG2 (see openzl_amd/build.py)
#define ZL_GCAT_(a, b) a##_##b
#define ZL_GCAT(a, b) ZL_GCAT_(a, b)
#define ZL_GNAME(f) ZL_GCAT(f, ZL_G)

// Wave issue priority of the sort / tail kernels: in a pipeline they share every SIMD with two waves of the accumulation kernel, which would
// otherwise win most issue slots (a 1-ms sort kernel then takes 5-9 ms); their own VALU demand is tiny.
#ifdef ZL_NO_SIDE_PRIO
#define ZL_SIDE_PRIO() ((void)0)
#else
#define ZL_SIDE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
#define ZL_CHUNK_MAX 64    // entries per lane in msm_accumulate (smaller for small inputs: more lanes, shorter chains)
#define ZL_BIG_SPAN 64     // buckets cut into more chunks than this are merged by a whole block ...
#define ZL_BIG_SPAN_SMALL 8  // ... 8 for small inputs: a lane folds its partials serially, and 64 dependent additions (1.2 ms for G1, 3 ms
                             // for G2) were the whole tail of a small Groth16 proof; for large inputs the serial fold is the cheaper one
#define ZL_GIANT_SPAN 4096 // ... and into more than this by ZL_GIANT_PARTS blocks (two stages)
#define ZL_GIANT_PARTS 32

__device__ __forceinline__ uint32_t zl_get_bits(const uint32_t* __restrict__ s, int pos, int c) {
    // bits [pos, pos+c) of a 256-bit little-endian integer (c <= 24); bits above 255 read as 0
    int word = pos >> 5, sh = pos & 31;
    if (word >= 8) return 0;
    uint64_t v = s[word];
    if (word + 1 < 8) v |= (uint64_t)s[word + 1] << 32;
    return (uint32_t)(v >> sh) & ((1u << c) - 1);
}

// The ABI takes canonical scalars (< r, what ark's into_repr() yields).  A scalar with bits at or above SC_BITS cannot be one; the window
// layout (W = ceil((SC_BITS + 1) / c) windows, spread top window) silently drops or misplaces such bits, so the recoder flags them and the
// call returns ZL_EINVAL instead of a wrong sum.
__device__ __forceinline__ void zl_flag_wide_scalar(uint32_t top_word, int sc_bits, uint32_t* __restrict__ bad) {
    if (bad && (top_word >> (sc_bits - 224)) != 0u) atomicOr(bad, 1u);
}
