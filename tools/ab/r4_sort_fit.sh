#!/bin/bash
# round 4: the kernels of the wide sort sized to fit beside a G2 accumulation (96 registers per SIMD left): part_scatter_st 104 -> 96 registers, fine_sort on 512-lane
# workgroups below 2^26 entries, fine_sort_big on 256-lane workgroups (tiles of 4096).  In-tree build against the previous commit (tools/libzl_prev.so), interleaved.
mkdir -p gpurun_out
out=gpurun_out/r04_sort_fit_ab.log
: > $out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for rep in 1 2 3; do
for lib in tools/libzl_prev.so openzl_amd/libzl_backend.so; do
  echo "== $lib" >> $out
  ZL_BACKEND_LIB=$PWD/$lib ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_BACKEND_LIB=$PWD/$lib PRE_LEGS=1 ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" | sed 's/^/PRE_LEGS /' >> $out
done
done
for lib in tools/libzl_prev.so openzl_amd/libzl_backend.so; do
  echo "== $lib: MSMs" >> $out
  ZL_BACKEND_LIB=$PWD/$lib BATCH=6 python tools/msm_sweep.py 20 22 24 2>&1 | grep "2^" >> $out
  ZL_BACKEND_LIB=$PWD/$lib PRE=20 python tools/msm_sweep.py 20 2>&1 | grep "2^" >> $out
  ZL_BACKEND_LIB=$PWD/$lib CURVE=bn254 ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
  ZL_BACKEND_LIB=$PWD/$lib python bench.py --no-cpu --no-configs --no-ntt --groth16-k 0 --fixed-key -1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench value ms', d['ms_per_step'], 'skewed ms', d['msm_skewed_scalars']['ms_per_step'], 'pcie', d['pcie_inclusive']['ms_per_msm'])" >> $out
done
cat $out
TL_MIN_US=150 bash tools/ab/r4_g16_tl.sh > /dev/null 2>&1
