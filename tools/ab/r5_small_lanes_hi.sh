for rep in 1 2; do
for cfg in "2 4" "0 4" "2 2" "2 3"; do
  set -- $cfg
  echo "SMALL_LANE_PRIO=$1 LANES=$2: $(ZL_TUNE_SMALL_LANE_PRIO=$1 ZL_TUNE_SIDE_LANES=$2 ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth | sed 's/Groth16 //' | tr '\n' ' ')"
done
done
for p in 2 0; do echo "SMALL_LANE_PRIO=$p msm batches: $(ZL_TUNE_SMALL_LANE_PRIO=$p BATCH=6 CS=0 python tools/msm_sweep.py 16 18 2>&1 | grep BATCH | tr '\n' ' ')"; done
