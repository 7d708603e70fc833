#!/bin/bash
# round 5: host-scalar MSM (zl_msm) with the shards on one carried bucket set against round 4's independent shard jobs (ZL_TUNE_HOST_CARRY=0), shard plans swept
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_host_carry_ab.log; : > $L
python -m pytest tests/test_gpu_msm.py -q -m gpu -x -k "carried or known_discrete_log_2_24 or adversarial" 2>&1 | tail -3 >> $L
for rep in 1 2; do
  for cfg in "0 -" "1 -" "1 19,21,23" "1 18,20,22" "1 20,22" "1 19,21" "1 20,22,23"; do
    set -- $cfg
    echo "== ZL_TUNE_HOST_CARRY=$1 ZL_TUNE_HOST_SHARDS=$2" >> $L
    if [ "$2" = "-" ]; then ZL_TUNE_HOST_CARRY=$1 python tools/host_msm.py 2>&1 | grep -i "ms" | tail -3 >> $L; else ZL_TUNE_HOST_CARRY=$1 ZL_TUNE_HOST_SHARDS=$2 python tools/host_msm.py 2>&1 | grep -i "ms" | tail -3 >> $L; fi
  done
done
cat $L
