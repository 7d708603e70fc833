#!/bin/bash
# round 6: the host pool with one job slot (libzl_backend.oldpool.so) against one job per concurrent caller
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_pool_ab.log; : > $L
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in oldpool new; do
  if [ $v = oldpool ]; then export ZL_BACKEND_LIB=$R/openzl_amd/libzl_backend.oldpool.so; else unset ZL_BACKEND_LIB; fi
  echo "== $v" >> $L
  for k in 1 8 64; do ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L; done
  CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
  ITERS=10 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
  python tools/msm_sweep.py 12 16 20 2>&1 | grep -v amdgpu.ids >> $L
done
done
unset ZL_BACKEND_LIB
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 2>&1 | grep -v "prove \|synth\|amdgpu.ids" | tail -19 > $O/r06_pool_trace_k1.txt
cat $L
