// zl_msm_acc.hip -- device code of the bucket accumulation kernels (zl_msm_accumulate.h) for one group: -DZL_G=BlsG1|BnG1|BlsG2|BnG2.
#include "zl_msm_accumulate.h"
#ifndef ZL_G
#error "compile with -DZL_G=<group config>"
#endif
ZL_MSM_ACCUMULATE_KERNELS(, ZL_G)
