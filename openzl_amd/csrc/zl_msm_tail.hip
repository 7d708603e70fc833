// zl_msm_tail.hip -- device code of the merge / scalar-1 / bucket-reduction kernels (zl_msm_reduce.h) for one group: -DZL_G=BlsG1|BnG1|BlsG2|BnG2.
#include "zl_msm_reduce.h"
#ifndef ZL_G
#error "compile with -DZL_G=<group config>"
#endif
ZL_MSM_TAIL_KERNELS(, ZL_G)
