#!/bin/bash
# round 6: the LDS row swizzle of the NTT tile (tile_pos; variant build: ZL_EXTRA_FLAGS=-DZL_NTT_ROW_SWIZZLE=1 ZL_BUILD_TAG=swz python -m openzl_amd.build) against round 5's layout (the product build).
# (When profiles/r06_ntt_swizzle_ab.log was taken the swizzle was the product default and the variant was the old layout: same two builds, names swapped.)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_ntt_swizzle_ab.log; : > $L
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ntt.py tests/test_gpu_sharded_ntt.py tests/test_groth16.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -2 >> $L
for rep in 1 2 3; do
  echo "== swizzled rows" >> $L; for ln in 24 20 22; do ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.swz.so python tools/ntt_one.py $ln 8 2>&1 | tail -1 >> $L; done
  echo "== round 5's layout (product)" >> $L; for ln in 24 20 22; do python tools/ntt_one.py $ln 8 2>&1 | tail -1 >> $L; done
done
C2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for v in product swz; do
  lib=$PWD/openzl_amd/libzl_backend.so; [ $v = swz ] && lib=$PWD/openzl_amd/libzl_backend.swz.so
  ZL_BACKEND_LIB=$lib timeout 600 rocprofv3 --pmc $C2 --kernel-trace -d $O/psq_$v -o s -f csv -- python tools/ntt_one.py 24 2 > $O/pmc_$v.log 2>&1
  cp $(find $O/psq_$v -name "*counter_collection.csv" | head -1) $O/ntt_sq_$v.csv 2>/dev/null
  rm -rf $O/psq_$v
done
ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.swz.so ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
cat $L
