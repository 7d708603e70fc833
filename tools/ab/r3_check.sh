#!/bin/bash
# gpu tests + the bench line (N = 1 default, bare 2-rank gloo launch) in one gpurun call
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3check}
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench rc=$?"
tail -c 600 $O/bench_n1.err
