#!/bin/bash
# round 5: chunk length of k_msm_accumulate chosen so that the waves fill the machine's 3 x 1024 wave slots a whole number of times (ZL_TUNE_CHUNK sweeps)
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_chunk_sweep.log; : > $L
run() { echo "== 2^$1 ZL_TUNE_CHUNK=$2 (0 = the library's choice)" >> $L; if [ "$2" = 0 ]; then BATCH=6 python tools/msm_sweep.py $1 2>&1 | grep "2^" >> $L; else ZL_TUNE_CHUNK=$2 BATCH=6 python tools/msm_sweep.py $1 2>&1 | grep "2^" >> $L; fi; }
for rep in 1 2; do
for c in 0 86 43 57; do run 20 $c; done
for c in 0 107 160 128; do run 22 $c; done
for c in 0 133 120 149; do run 24 $c; done
for c in 0 44 22; do run 18 $c; done
done
cat $L
