for rep in 1 2; do
for pre in 0 1; do
for cfg in "0 4" "2 3" "3 3"; do
  set -- $cfg
  for k in 1 64; do
    echo "PRE_LEGS=$pre SMALL_LANE_PRIO=$1 LANES=$2: $( ( [ $pre = 1 ] && export PRE_LEGS=1; ZL_TUNE_SMALL_LANE_PRIO=$1 ZL_TUNE_SMALL_LANES=$2 ITERS=40 python tools/g16_one.py $k 2>&1 | tail -1 ) )"
  done
done
done
done
