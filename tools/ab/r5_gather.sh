#!/bin/bash
# round 5: what the 128-B base gathers cost k_msm_accumulate in time and in clock: the clock-reading build with every base index folded into the first 2^b points
# (ZL_TUNE_ACC_CLK_IDX_BITS=b: wrong sums, same arithmetic, gathers served from L2 / MALL) against the product's gather, interleaved
O=gpurun_out/r5; mkdir -p $O; L=$O/r05_acc_gather_ab.log; : > $L
for rep in 1 2; do
  for b in 31 20 14; do
    echo "== ZL_TUNE_ACC_CLK_IDX_BITS=$b (31 = the product's gather over 2^24 points = 2 GiB; 20 = 128 MiB: MALL; 14 = 2 MiB: L2)" >> $L
    ZL_TUNE_ACC_CLK_IDX_BITS=$b python tools/clock_probe.py 24 2 2>&1 | grep "^accumulate" | cut -c1-520 >> $L
  done
done
cat $L
python -m pytest tests/test_gpu_bench_smoke.py -q -m gpu -x -k "one_rank or sharded_helpers" 2>&1 | tail -15
python bench.py --no-configs --groth16-k 0 --no-skew --fixed-key -1 --no-cpu > $O/bench_quick.json 2> $O/bench_quick.err; tail -3 $O/bench_quick.err; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5/bench_quick.json") if l.startswith("{")][-1])
print(json.dumps(d["roofline"]["int_alu"], indent=1)[:2500])
print(json.dumps(d["ntt"]["roofline"], indent=1)[:1500])
print(d["ms_per_step"], d["pcie_inclusive"])
PY
