#!/bin/bash
# round 4: large proof x PRE_LEGS after moving the copy stream out of the default priority class
mkdir -p gpurun_out
out=gpurun_out/r04_g16_pre2.log
: > $out
for rep in 1 2 3; do
for pre in 0 1; do
  if [ $pre = 1 ]; then export PRE_LEGS=1; else unset PRE_LEGS; fi
  echo "== 958465 constraints, PRE_LEGS=$pre" >> $out
  ITERS=12 python tools/g16_one.py 4096 2>&1 | grep "prove k=" >> $out
done
done
unset PRE_LEGS
python tools/host_msm.py 24 6 2>&1 | grep host >> $out
python bench.py --no-cpu --no-configs --no-skew --fixed-key -1 2>/dev/null | tail -1 > gpurun_out/bench_g16.json
python - <<'PY' >> $out
import json
d=json.loads(open('gpurun_out/bench_g16.json').read())
g=d['groth16']; print('bench (headline+pcie+ntt+g16):', {k:g[k] for k in g if k.startswith('prove')}); print([(x['hashes'], x['prove_ms']) for x in g['small_circuits']]); print('pcie', d['pcie_inclusive']['ms_per_msm'], 'value ms', d['ms_per_step'], 'ntt', d['ntt']['forward_ms'])
PY
cat $out
