#!/bin/bash
# round 5: lazy NTT passes with 2048-element tiles on 512-lane workgroups (ZL_EXTRA_FLAGS="-DNTT28_TILE_LOG=11 -DNTT28_THREADS=512" ZL_BUILD_TAG=t2048) against the 1024-element tiles of the product build
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_ntt_tile_ab.log; : > $L
ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.t2048.so python -m pytest tests/test_gpu_ntt.py -q -m gpu -x 2>&1 | tail -2 >> $L
for rep in 1 2 3; do
  echo "== tile 1024 x 256 lanes (product)" >> $L; python tools/ntt_one.py 24 8 2>&1 | tail -2 >> $L; python tools/ntt_one.py 20 8 2>&1 | tail -1 >> $L; python tools/ntt_one.py 22 8 2>&1 | tail -1 >> $L
  echo "== tile 2048 x 512 lanes" >> $L; ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.t2048.so python tools/ntt_one.py 24 8 2>&1 | tail -2 >> $L; ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.t2048.so python tools/ntt_one.py 20 8 2>&1 | tail -1 >> $L;  ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.t2048.so python tools/ntt_one.py 22 8 2>&1 | tail -1 >> $L
done
cat $L
