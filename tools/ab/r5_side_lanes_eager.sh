#!/bin/bash
# with every stream created at zl_ctx_create: how many lanes for the four large G1 jobs of the 958 465-constraint proof (4 = one of them shares a hardware queue with the G2 MSM's stream)
for rep in 1 2 3; do for l in 4 3 2; do
  echo "SIDE_LANES=$l: $(ZL_TUNE_SIDE_LANES=$l ITERS=11 python tools/g16_one.py 4096 2>&1 | tail -1)   bn254: $(CURVE=bn254 ZL_TUNE_SIDE_LANES=$l ITERS=11 python tools/g16_one.py 4096 2>&1 | tail -1)"
done; done
