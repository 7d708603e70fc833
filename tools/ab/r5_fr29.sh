#!/bin/bash
# round 5: the NTT passes on 9 x 29-bit limbs (product build) against round 4's 10 x 28-bit limbs (ZL_EXTRA_FLAGS=-DZL_NTT_FR28 ZL_BUILD_TAG=fr28), interleaved on one box
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_ntt_fr29_ab.log; : > $L
python -m pytest tests/test_fr28.py tests/test_gpu_ntt.py tests/test_gpu_field_kat.py tests/test_gpu_sharded_ntt.py tests/test_groth16.py -q -m gpu -x 2>&1 | tail -4 >> $L
for rep in 1 2 3; do
  for lib in openzl_amd/libzl_backend.fr28.so openzl_amd/libzl_backend.so; do
    echo "== $lib" >> $L
    ZL_BACKEND_LIB=$PWD/$lib python tools/ntt_one.py 24 8 2>&1 | tail -3 >> $L
    ZL_BACKEND_LIB=$PWD/$lib python tools/ntt_one.py 20 8 2>&1 | tail -1 >> $L
  done
done
cat $L
