#!/bin/bash
# Round 5 re-sweep of the window-table widths of the large proving key (round 3: G1 19 / 21 worse than 20, G2 17 / 18 / 20 = 19.4 / 19.2 / 20.5 against 19.4 at 16), interleaved on one box.
for rep in 1 2; do
for cfg in "20 16" "20 17" "20 18" "20 19" "21 16" "19 16" "19 18"; do
  set -- $cfg
  echo "G1 c=$1 G2 c=$2: $(ITERS=9 ZL_TUNE_G1_TABLE_C=$1 ZL_TUNE_G2_TABLE_C=$2 python tools/g16_one.py 4096 2>&1 | tail -1)"
done
done
