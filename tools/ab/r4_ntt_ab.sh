#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r04_ntt_lazy_ab.log; : > $out
python -m pytest tests/test_fr28.py tests/test_gpu_ntt.py tests/test_gpu_sharded_ntt.py -m gpu -x -q 2>&1 | tail -5 >> $out
for mode in lazy nolazy lazy nolazy; do
  if [ $mode = nolazy ]; then export ZL_NTT_NO_LAZY=1; else unset ZL_NTT_NO_LAZY; fi
  echo "== $mode" >> $out
  python tools/ntt_one.py 24 24 2>&1 | tail -6 >> $out
  python tools/ntt_one.py 20 40 2>&1 | tail -2 >> $out
done
unset ZL_NTT_NO_LAZY
cat $out
python -m pytest tests/test_groth16.py tests/test_golden_vectors.py -m gpu -x -q 2>&1 | tail -3
