#!/bin/bash
# creation order of a ctx's streams (zl_ctx_streams_init) = which kernel chains share a hardware queue: proofs of three sizes + BN254, interleaved, three passes
for rep in 1 2 3; do for o in 0 1 2 3 4; do
  echo "ORDER=$o: k=4096 $(ZL_TUNE_STREAM_ORDER=$o ITERS=11 python tools/g16_one.py 4096 2>&1 | tail -1 | sed 's/prove k=4096: //')  | k=64 $(ZL_TUNE_STREAM_ORDER=$o ITERS=40 python tools/g16_one.py 64 2>&1 | tail -1 | sed 's/prove k=64: //') | k=1 $(ZL_TUNE_STREAM_ORDER=$o ITERS=40 python tools/g16_one.py 1 2>&1 | tail -1 | sed 's/prove k=1: //')"
done; done
