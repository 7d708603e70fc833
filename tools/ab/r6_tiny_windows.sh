#!/bin/bash
O=gpurun_out/r6; mkdir -p $O; L=$O/r06_tiny_windows.log; : > $L
for rep in 1 2; do
echo "== G2 2^8 (first line: a warm-up of the process, second: the picker)" >> $L
CS=0,0,4,5,6,7,8,9,10 python tools/msm_sweep.py --g2 8 2>&1 | grep -v amdgpu.ids >> $L
echo "== G1 2^8, 2^10" >> $L
CS=0,0,5,6,7,8,9,10,11 python tools/msm_sweep.py 8 10 2>&1 | grep -v amdgpu.ids >> $L
done
python bench.py > gpurun_out/r6/r06_bench_final_boxB.json 2> gpurun_out/r6/bench_boxB.err
cat $L
