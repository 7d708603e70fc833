#!/bin/bash
# round 4: phase-major issue of the four G1 jobs of a large proof: on / off, interleaved on one box
mkdir -p gpurun_out
out=gpurun_out/r04_g16_phased_ab.log
: > $out
python -m pytest tests/test_groth16.py tests/test_host_mirror.py tests/test_golden_vectors.py tests/test_gpu_msm.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" >> $out
for mode in phased jobmajor phased jobmajor phased jobmajor; do
  if [ $mode = jobmajor ]; then export ZL_TUNE_PHASED_MIN_LOG=40; else unset ZL_TUNE_PHASED_MIN_LOG; fi
  echo "== 958465 constraints, $mode" >> $out
  ITERS=14 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
for k in 512 2048; do
for mode in phased jobmajor; do
  if [ $mode = jobmajor ]; then export ZL_TUNE_PHASED_MIN_LOG=40; else export ZL_TUNE_PHASED_MIN_LOG=10; fi
  echo "== k=$k, $mode (threshold forced)" >> $out
  ITERS=20 python tools/g16_one.py $k 2>&1 | grep "prove k=" >> $out
done
done
unset ZL_TUNE_PHASED_MIN_LOG
cat $out
python bench.py --no-cpu --no-configs --no-ntt --no-skew --fixed-key -1 2>/dev/null | tail -1 > gpurun_out/bench_g16.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_g16.json').read())
g=d['groth16']; print({k:g[k] for k in g if k.startswith('prove')}); print(g['small_circuits'])
PY
