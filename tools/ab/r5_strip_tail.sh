#!/bin/bash
# round 5: inline-asm pads stripped from the accumulation AND tail units (ZL_STRIP_ASM_NOPS=1 ZL_STRIP_UNITS=zl_msm_acc_,zl_msm_tail_ ZL_BUILD_TAG=strip) against the product build:
# the latency-bound sizes (a lone wave pays ~3.5 cycles per pad, tools/ubench2.hip)
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_strip_tail_ab.log; : > $L
ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.strip.so python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_groth16.py -q -m gpu -x 2>&1 | tail -2 >> $L
for rep in 1 2 3; do
  for lib in openzl_amd/libzl_backend.so openzl_amd/libzl_backend.strip.so; do
    echo "== $lib" >> $L
    ZL_BACKEND_LIB=$PWD/$lib SIZES=16,18,20 python tools/small_lat.py msm g16 2>&1 | grep -v "^$\|amdgpu.ids" | tail -12 >> $L
  done
done
cat $L
