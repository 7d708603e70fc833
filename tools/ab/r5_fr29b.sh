#!/bin/bash
# round 5: NTT passes on 9 x 29-bit limbs with 36-byte scratch slots; butterfly roots from LDS (default) or from global memory (ZL_TUNE_NTT_ROOTS_GLOBAL=1: four workgroups per CU),
# against round 4's 10 x 28-bit build (libzl_backend.fr28.so), interleaved on one box
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_ntt_fr29_ab2.log; : > $L
python -m pytest tests/test_gpu_ntt.py tests/test_gpu_sharded_ntt.py tests/test_groth16.py -q -m gpu -x 2>&1 | tail -2 >> $L
ZL_TUNE_NTT_ROOTS_GLOBAL=1 python -m pytest tests/test_gpu_ntt.py -q -m gpu -x 2>&1 | tail -2 >> $L
for rep in 1 2 3; do
  echo "== fr28 (round 4)" >> $L; ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.fr28.so python tools/ntt_one.py 24 8 2>&1 | tail -2 >> $L
  echo "== fr29, roots in LDS" >> $L; python tools/ntt_one.py 24 8 2>&1 | tail -2 >> $L; python tools/ntt_one.py 20 8 2>&1 | tail -1 >> $L
  echo "== fr29, roots in global memory" >> $L; ZL_TUNE_NTT_ROOTS_GLOBAL=1 python tools/ntt_one.py 24 8 2>&1 | tail -2 >> $L; ZL_TUNE_NTT_ROOTS_GLOBAL=1 python tools/ntt_one.py 20 8 2>&1 | tail -1 >> $L
done
cat $L
