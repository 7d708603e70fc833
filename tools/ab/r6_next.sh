#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/next
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_lanes.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > $O/tests.log
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt
python bench.py > $O/r06_bench_next.json 2> $O/bench.err
C="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/psq -o s -f csv -- python tools/ntt_one.py 24 2 > $O/pmc_sq_ntt.log 2>&1
cp $(find $O/psq -name "*counter_collection.csv" | head -1) $O/ntt_sq_lds.csv 2>/dev/null
C2="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"
timeout 600 rocprofv3 --pmc $C2 --kernel-trace -d $O/psq2 -o s -f csv -- python tools/ntt_one.py 24 2 > $O/pmc_sq_ntt2.log 2>&1
cp $(find $O/psq2 -name "*counter_collection.csv" | head -1) $O/ntt_sq_lds2.csv 2>/dev/null
rm -rf $O/psq $O/psq2
cat $O/tests.log; python tools/bench_digest.py $O/r06_bench_next.json 2>/dev/null | tail -2; wc -l $O/sq_counters.txt; tail -3 $O/pmc_sq_ntt.log
