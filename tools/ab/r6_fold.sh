#!/bin/bash
# round 6: small proofs with the folded C query + the key's fixed-base tables (ZL_TUNE_G16_FOLD_LOG_N = largest domain that folds; 0 = four G1 MSMs as before, tables only)
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_fold_ab.log; : > $L
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $L
timeout 1500 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py tests/test_gpu_multi.py tests/test_gpu_determinism.py tests/test_abi.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $L
ZL_TUNE_G16_FOLD_LOG_N=20 timeout 900 python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 >> $L
for rep in 1 2 3; do
for v in 0 12 16; do
  echo "== ZL_TUNE_G16_FOLD_LOG_N=$v" >> $L
  for k in 1 8 64; do
  ZL_TUNE_G16_FOLD_LOG_N=$v ITERS=40 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $L
  done
  ZL_TUNE_G16_FOLD_LOG_N=$v CURVE=bn254 ITERS=40 python tools/g16_one.py 1 2>&1 | grep "prove k=" >> $L
done
done
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 > $O/r06_fold_trace_k1.log 2>&1
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 64 > $O/r06_fold_trace_k64.log 2>&1
ZL_TUNE_G16_FOLD_LOG_N=16 ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 64 > $O/r06_fold_trace_k64_fold.log 2>&1
cat $L
