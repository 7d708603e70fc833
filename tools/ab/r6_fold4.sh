#!/bin/bash
# round 6: folded small proofs after (a) B assembled on the G2 worker, (b) no host wait for z
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_fold4.log; : > $L
timeout 1500 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py tests/test_gpu_multi.py tests/test_gpu_determinism.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $L
for rep in 1 2 3; do
  for k in 1 8 64; do
  ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
  CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" | sed 's/^/bn254 /' >> $L
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $L
done
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 > $O/r06_fold4_trace_k1.log 2>&1
cat $L
