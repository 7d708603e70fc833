// zl_msm_sort.h -- the curve-independent kernels of the MSM: signed-digit recoding, the (window, bucket) counting sorts and the scans
// (definitions: zl_msm_sort.hip, compiled ONCE; launched from MsmJob<G>::sort in zl_msm_job.h).
#pragma once
#include "zl_msm_common.h"

#define ZL_PT 4096      // entries of a staged partition tile (k_msm_part_scatter_st / k_msm_sub_scatter_st)
#define ZL_BT 16384     // entries of a tile of an oversized sub-group (k_msm_fine_sort_big)
#define SCAN_ITEMS 16
#define SCAN_BLOCK 256

// the eight per-job counter words in ONE launch (a 32-byte hipMemsetAsync at a 4-byte-aligned address runs as three fill kernels: 14 us of a small job's chain)
__global__ void __launch_bounds__(64) k_msm_zero_words(uint32_t* __restrict__ p, uint32_t n);
template <int MODE>
__global__ void __launch_bounds__(256) k_msm_digits(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W,
                                                    uint32_t* __restrict__ counters, uint32_t* __restrict__ entries,
                                                    uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                    int sc_bits, uint32_t* __restrict__ bad);
__global__ void __launch_bounds__(256) k_msm_recode(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, int spread_t, int glv, uint16_t* __restrict__ digits,
                                                             uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                             int sc_bits, uint32_t* __restrict__ bad);
__global__ void __launch_bounds__(1024) k_msm_hist_lds(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t per_slice, uint32_t NB,
                                                        uint32_t* __restrict__ counts);
__global__ void __launch_bounds__(256) k_msm_slice_prefix(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ tot);
__global__ void __launch_bounds__(1024) k_msm_scatter_range(const uint16_t* __restrict__ digits, uint32_t n, uint32_t H, uint32_t RB,
                                                                     const uint32_t* __restrict__ offsets, uint32_t* __restrict__ entries,
                                                                     const uint32_t* __restrict__ slice_prefix, uint32_t NB, uint32_t nslices, uint32_t per_slice,
                                                                     uint32_t parts, uint32_t W);
__global__ void __launch_bounds__(256) k_msm_recode_wide(const uint32_t* __restrict__ scalars, uint32_t n, int c, int W, uint32_t gw, int spread_t, int glv,
                                                                  uint16_t* __restrict__ lo16, uint8_t* __restrict__ hi8,
                                                                  uint32_t* __restrict__ ones_list, uint32_t* __restrict__ ones_count, const uint8_t* __restrict__ inf,
                                                                  int sc_bits, uint32_t* __restrict__ bad);
__global__ void __launch_bounds__(256) k_msm_part_hist(const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W, uint32_t G, uint32_t per_slice,
                                                                uint32_t nslices, uint32_t* __restrict__ counts);
__global__ void __launch_bounds__(256) k_msm_sub_hist(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_off, uint32_t G,
                                                               uint32_t stride, const uint32_t* __restrict__ total, uint32_t fslices,
                                                               uint32_t* __restrict__ counts);
__global__ void __launch_bounds__(256) k_msm_part_scatter_st(const uint16_t* __restrict__ lo16, const uint8_t* __restrict__ hi8, uint32_t n, uint32_t W,
                                                                      uint32_t G, uint32_t per_slice, uint32_t nslices, const uint32_t* __restrict__ part_off,
                                                                      uint32_t table_stride, uint32_t first, uint16_t* __restrict__ out_lo,
                                                                      uint32_t* __restrict__ out_idx);
__global__ void __launch_bounds__(256) k_msm_sub_scatter_st(const uint16_t* __restrict__ part_lo, const uint32_t* __restrict__ part_idx,
                                                                     const uint32_t* __restrict__ part_off, uint32_t G, uint32_t stride,
                                                                     const uint32_t* __restrict__ total, uint32_t fslices, const uint32_t* __restrict__ sub_off,
                                                                     uint16_t* __restrict__ out_lo, uint32_t* __restrict__ out_idx);
__global__ void __launch_bounds__(256) k_msm_fine_hist(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ sub_off, uint32_t SG,
                                                                uint32_t fslices, const uint32_t* __restrict__ total, uint32_t* __restrict__ counts);
__global__ void __launch_bounds__(1024) k_msm_fine_sort(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
                                                                const uint32_t* __restrict__ sub_off, uint32_t SG, uint32_t fslices,
                                                                const uint32_t* __restrict__ total, const uint32_t* __restrict__ offsets, uint32_t cap,
                                                                uint32_t* __restrict__ entries, unsigned long long* __restrict__ big_head,
                                                                uint32_t* __restrict__ big_items);
__global__ void __launch_bounds__(1024) k_msm_fine_sort_big(const uint16_t* __restrict__ lo2, const uint32_t* __restrict__ idx2,
                                                                    const uint32_t* __restrict__ sub_off, uint32_t SG, uint32_t fslices,
                                                                    const uint32_t* __restrict__ total, const unsigned long long* __restrict__ big_head,
                                                                    const uint32_t* __restrict__ big_items, uint32_t* __restrict__ cursor,
                                                                    uint32_t* __restrict__ entries);
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_block_sums(const uint32_t* __restrict__ in, uint32_t count, uint32_t* __restrict__ block_sums);
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t* __restrict__ total_out,
                                                          const uint32_t* __restrict__ flag_in = nullptr /* copied to total_out[1] */);
__global__ void __launch_bounds__(1024) k_msm_prefix_small(uint32_t* __restrict__ counts, uint32_t NB, uint32_t nslices, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ cursor, const uint32_t* __restrict__ flag_in);
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* __restrict__ in, uint32_t count, const uint32_t* __restrict__ block_sums,
                                                           uint32_t* __restrict__ out, uint32_t* __restrict__ out2);
