#!/bin/bash
# round 5: the product build (inline-asm pads stripped from the accumulation units, build.py) against hipcc's own output (ZL_KEEP_ASM_NOPS=1 ZL_BUILD_TAG=nops), interleaved on one box
#   gpurun --timeout 1200 -- 'bash tools/ab/r5_nop_ab.sh'
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_nop_ab.log; : > $L
python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_groth16.py -q -m gpu -x 2>&1 | tail -3 >> $L
for rep in 1 2; do
  for lib in openzl_amd/libzl_backend.nops.so openzl_amd/libzl_backend.so; do
    echo "== $lib" >> $L
    ZL_BACKEND_LIB=$PWD/$lib python tools/clock_probe.py 24 2 2>&1 | grep "^accumulate" | cut -c1-420 >> $L
    ZL_BACKEND_LIB=$PWD/$lib BATCH=6 python tools/msm_sweep.py 16 20 24 2>&1 | grep "2^" >> $L
    ZL_BACKEND_LIB=$PWD/$lib python tools/msm_sweep.py --g2 16 20 2>&1 | grep "2^" >> $L
    ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 python tools/msm_sweep.py 20 24 2>&1 | grep "2^" >> $L
  done
done
cat $L
