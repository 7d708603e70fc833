#!/bin/bash
# One gpurun call that regenerates every measured artifact of a round under gpurun_out/final/ (copy into profiles/ afterwards).
#   gpurun --timeout 2400 -- 'bash tools/final_artifacts.sh'
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
PMCARGS="--steps 1 --warmup 0 --no-cpu --no-ntt --no-skew --groth16-k 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -f csv -- python bench.py $PMCARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -f csv -- python bench.py $PMCARGS > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) 22 > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/r01_pmc_traffic_final.json
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.txt
rocprofv3 --kernel-trace --stats -d $O/prof_np -o bench -- python bench.py --no-pipeline --no-ntt --no-skew --no-cpu --groth16-k 0 > $O/bench_unpipelined.json 2> $O/bench_unpipelined.err
python tools/prof_summary.py $(find $O/prof_np -name "*.db" | head -1) > $O/kernel_stats_unpipelined.txt
python bench.py > $O/bench.json 2> $O/bench.err
ZL_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --log-n 20 --ntt-log-n 20 > $O/bench_2rank_gloo_1gpu.json 2> $O/bench_2rank.err
(./tools/fbench28; ./tools/fbench28r; ./tools/fbench_bfly) > $O/fbench_field_mul.log 2>&1
python tools/msm_sweep.py 16 18 20 22 24 > $O/msm_sweep_plain.log 2>&1
python tools/msm_sweep.py --g2 16 20 > $O/msm_sweep_g2.log 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/prof $O/prof_np
ls -la $O
