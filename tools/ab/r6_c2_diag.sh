#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r06_c2_diag.log
: > $out
timeout 900 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_msm_g2.py tests/test_groth16.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5 >> $out
for pre in 0 1 0 1; do echo "== PRE=$pre" >> $out; PRE=$pre python tools/ab/r6_c2_diag.py 2>&1 | grep -v amdgpu.ids >> $out; done
cat $out
