#!/bin/bash
# round 6: config 2 inside bench.py at several points of the run
mkdir -p gpurun_out
out=gpurun_out/r06_c2_bench2.log
: > $out
for rep in 1 2; do
    ZL_BENCH_DBG_C2=1 python bench.py --steps 6 --no-cpu --no-ntt --groth16-k 0 --no-pcie --no-live-traffic --fixed-key -1 --no-skew 2>&1 >/dev/null | grep "dbg c2" >> $out
done
cat $out
