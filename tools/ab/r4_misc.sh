#!/bin/bash
mkdir -p gpurun_out
./tools/fbench_f64 > gpurun_out/r04_fbench_f64.log 2>&1; grep -E "gate|Fr " gpurun_out/r04_fbench_f64.log
out=gpurun_out/r04_host_shards.log; : > $out
python tools/host_msm.py 24 6 >> $out 2>&1
for plan in "20,22" "21,22" "20,21,22" "20,22,23" "21"; do ZL_TUNE_HOST_SHARDS=$plan python tools/host_msm.py 24 6 2>&1 | grep host >> $out; done
cat $out
python tools/ntt_one.py 24 25 > gpurun_out/r04_ntt_drift.log 2>&1; cat gpurun_out/r04_ntt_drift.log | awk 'NR%3==1'
python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_g2.py tests/test_gpu_msm_fuzz.py tests/test_groth16.py -m gpu -x -q 2>&1 | tail -3
