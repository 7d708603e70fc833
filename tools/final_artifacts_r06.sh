#!/bin/bash
# One gpurun call that regenerates the measured artifacts of round 6 under gpurun_out/final/ (copy into profiles/ afterwards).
#   gpurun --timeout 2400 -- 'bash tools/final_artifacts_r06.sh [pmc]'
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
csvof() { find $1 -name "*counter_collection.csv" | head -1; }
dbof() { find $1 -name "*.db" | head -1; }
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r06_gpu_tests.log
# ---- HBM traffic (PMC; separate passes per counter, as the guide prescribes) -------------------------------------------------------
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o f -f csv -- python tools/msm_one.py 24 0 -1 1 > $O/pmc_msm_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o w -f csv -- python tools/msm_one.py 24 0 -1 1 > $O/pmc_msm_write.log 2>&1
CW=$(grep -o "c=[0-9]*" $O/pmc_msm_fetch.log | tail -1 | cut -d= -f2)   # the window the library picked for the plain path
python tools/pmc_fold.py msm $(csvof $O/pf) $(csvof $O/pw) 24 $CW 0 > $O/r06_pmc_traffic.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf2 -o f -f csv -- python tools/msm_one.py 24 0 22 1 > $O/pmc_msmt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw2 -o w -f csv -- python tools/msm_one.py 24 0 22 1 > $O/pmc_msmt_write.log 2>&1
python tools/pmc_fold.py msm $(csvof $O/pf2) $(csvof $O/pw2) 24 22 1 > $O/r06_pmc_traffic_fixed_key.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf3 -o f -f csv -- python tools/ntt_one.py 24 2 > $O/pmc_ntt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw3 -o w -f csv -- python tools/ntt_one.py 24 2 > $O/pmc_ntt_write.log 2>&1
python tools/pmc_fold.py ntt $(csvof $O/pf3) $(csvof $O/pw3) 24 > $O/r06_pmc_traffic_ntt.json
mkdir -p profiles; cp $O/r06_pmc_traffic.json $O/r06_pmc_traffic_ntt.json profiles/
if [ "${1:-all}" = "pmc" ]; then rm -rf $O/pf $O/pw $O/pf2 $O/pw2 $O/pf3 $O/pw3; cat $O/r06_pmc_traffic.json | head -c 600; exit 0; fi
# ---- kernel stats ------------------------------------------------------------------------------------------------------------------
rocprofv3 --kernel-trace --stats -d $O/p1 -o t -- python bench.py > $O/r06_bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python tools/prof_summary.py $(dbof $O/p1) > $O/r06_kernel_stats_bench_default.txt
# the headline leg alone (no fixed-key / skewed / NTT / Groth16 / CPU / configs legs): every big k_msm_accumulate launch in this trace is one timed or warm-up step of `value`,
# so the table's big_avg_us is directly comparable with roofline.kernel_ms of the JSON line written by the same command
rocprofv3 --kernel-trace --stats -d $O/p7 -o t -- python bench.py --no-skew --fixed-key -1 --no-ntt --groth16-k 0 --no-cpu --no-configs --no-pcie > $O/r06_bench_headline_under_rocprof.json 2> $O/bench_headline_under_rocprof.err
python tools/prof_summary.py $(dbof $O/p7) > $O/r06_kernel_stats_bench_headline.txt
rocprofv3 --kernel-trace --stats -d $O/p2 -o t -- python tools/msm_one.py 24 0 -1 3 > $O/msm_plain.log 2>&1
python tools/prof_summary.py $(dbof $O/p2) reduce_tree > $O/r06_kernel_stats_msm_plain_single_call.txt
rocprofv3 --kernel-trace --stats -d $O/p3 -o t -- python tools/msm_one.py 24 0 22 3 > $O/msm_table.log 2>&1
python tools/prof_summary.py $(dbof $O/p3) > $O/r06_kernel_stats_msm_fixed_key_single_call.txt
rocprofv3 --kernel-trace --stats -d $O/p4 -o t -- python tools/ntt_one.py 24 5 > $O/ntt_one.log 2>&1
python tools/prof_summary.py $(dbof $O/p4) k_ntt_pass > $O/r06_kernel_stats_ntt_2_24.txt
rocprofv3 --kernel-trace --stats -d $O/p5 -o t -- python tools/g16_one.py > $O/g16_one.log 2>&1
python tools/prof_summary.py $(dbof $O/p5) > $O/r06_kernel_stats_groth16.txt
python tools/timeline.py $(dbof $O/p5) 1500 1 200 > $O/r06_g16_timeline.txt 2>&1   # last proof: launches >= 200 us with their queues
rocprofv3 --kernel-trace --stats -d $O/p6 -o t -- python tools/msm_sweep.py --g2 20 > $O/g2_sweep.log 2>&1
python tools/prof_summary.py $(dbof $O/p6) > $O/r06_kernel_stats_g2_2_20.txt
# ---- bench lines and sweeps --------------------------------------------------------------------------------------------------------
python bench.py > $O/r06_bench_final.json 2> $O/bench.err
ZL_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --log-n 20 --ntt-log-n 20 --groth16-k 64 --config4-log-total 22 --strong-log-total 20 > $O/r06_bench_2rank_gloo_1gpu.json 2> $O/bench_2rank.err
python bench.py --transport mctx --gpus 4 --mctx-devices 0,0,0,0 --log-n 20 --ntt-log-n 18 --steps 3 --warmup 1 > $O/r06_bench_mctx_4_virtual_ranks.json 2> $O/bench_mctx.err
BATCH=6 CS=16,18,19,20 python tools/msm_sweep.py 16 18 20 22 24 > $O/r06_msm_sweep_plain.log 2>&1
PRE=20,22 BATCH=6 python tools/msm_sweep.py 24 > $O/r06_msm_sweep_fixed_key.log 2>&1
python tools/msm_sweep.py --g2 12 14 16 18 20 22 > $O/r06_msm_sweep_g2.log 2>&1
CURVE=bn254 python tools/msm_sweep.py --g2 16 20 >> $O/r06_msm_sweep_g2.log 2>&1
for ln in 16 20; do
  rocprofv3 --kernel-trace --stats -d $O/ps$ln -o t -- python tools/msm_one.py $ln 0 -1 4 > $O/msm_one_$ln.log 2>&1
  python tools/timeline.py $(dbof $O/ps$ln) 150 > $O/r06_timeline_msm_2_$ln.txt 2>&1
done
for k in 1 64; do
  rocprofv3 --kernel-trace --stats -d $O/pg$k -o t -- python tools/g16_one.py $k > $O/g16_one_$k.log 2>&1
  python tools/timeline.py $(dbof $O/pg$k) 150 > $O/r06_timeline_g16_k$k.txt 2>&1
done
python tools/small_lat.py > $O/r06_small_latency.log 2>&1
# host-side phase times of a one-hash proof: folded form (default), the four-MSM form, BN254
ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 2>&1 | grep -v "prove \|synth\|amdgpu.ids" | tail -19 > $O/r06_fold_trace_after.txt
ZL_TUNE_G16_FOLD_LOG_N=0 ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 2>&1 | grep -v "prove \|synth\|amdgpu.ids" | tail -30 > $O/r06_fold_trace_fourjobs.txt
CURVE=bn254 ZL_HOST_TRACE=1 ITERS=6 python tools/g16_one.py 1 2>&1 | grep -v "prove \|synth\|amdgpu.ids" | tail -19 > $O/r06_fold_trace_after_bn254.txt
# effective clocks (in-kernel s_memtime / s_memrealtime) and the SQ issue / stall counters of the two dominant kernels
ZL_BACKEND_LIB=$R/openzl_amd/libzl_backend.measure.so python tools/clock_probe.py 24 3 > $O/r06_clock_probe_2_24.log 2>&1   # the clock-reading accumulation exists in -DZL_MEASURE builds only (ZL_EXTRA_FLAGS=-DZL_MEASURE ZL_BUILD_TAG=measure python -m openzl_amd.build)
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
for what in "msm_2_24 tools/msm_one.py 24 0 -1 1" "ntt_2_24 tools/ntt_one.py 24 2" "g2_2_20 tools/msm_sweep.py --g2 20"; do
  set -- $what; name=$1; shift
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/psq_$name -o s -f csv -- python "$@" > $O/pmc_sq_$name.log 2>&1
  python tools/pmc_sq.py $(find $O/psq_$name -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $C --kernel-trace -f csv -- python $*" > $O/r06_pmc_sq_$name.json
  rm -rf $O/psq_$name
done
rocprofv3 --kernel-trace --stats -d $O/pb -o t -- python tools/batch_trace.py 24 5 > $O/batch_trace.log 2>&1
python tools/timeline.py $(dbof $O/pb) 2000 1 150 > $O/r06_timeline_batch_2_24.txt 2>&1
rm -rf $O/ps16 $O/ps20 $O/pg1 $O/pg64 $O/pb
CURVE=bn254 BATCH=6 CS=16,19,20 python tools/msm_sweep.py 20 24 > $O/r06_msm_sweep_bn254.log 2>&1
G16_WIRE=1 python tools/g16_one.py 2>&1 | grep -v amdgpu.ids > $O/r06_g16_key_wire.log
rm -rf $O/pf $O/pw $O/pf2 $O/pw2 $O/pf3 $O/pw3 $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7
ls -la $O
