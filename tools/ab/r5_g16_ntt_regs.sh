#!/bin/bash
# round 5: does the nine-limb NTT pass still fit beside the 416-register G2 accumulation of a proof?  product build (<= 96 registers: NTT28_MIN_BLOCKS = 5) against
# the uncapped build (-DNTT28_MIN_BLOCKS=2: 101-104 registers) and round 4's ten-limb passes (-DZL_NTT_FR28: 94 registers), interleaved on one box
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_g16_ntt_regs_ab.log; : > $L
for rep in 1 2 3; do
  for v in so lb2.so fr28.so; do
    echo "== libzl_backend.$v" >> $L
    ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.$v python tools/ntt_one.py 24 6 2>&1 | tail -1 >> $L
    ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.$v ITERS=14 python tools/g16_one.py 2>&1 | tail -1 >> $L
    ZL_BACKEND_LIB=$PWD/openzl_amd/libzl_backend.$v ITERS=14 CURVE=bn254 python tools/g16_one.py 2>&1 | tail -1 >> $L
  done
done
cat $L
