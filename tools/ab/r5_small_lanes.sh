#!/bin/bash
# Round 5: the lanes of SMALL side-by-side MSM jobs in the low stream-priority class (a hardware-queue pool of their own) against the shared default-class lanes,
# at the runtime's default of 4 hardware queues per pool and at 8; small proofs and small MSM batches.  Interleaved, two passes.
for rep in 1 2; do
for cfg in "1 4" "0 4" "1 2" "0 2"; do
  set -- $cfg
  echo "SMALL_LANE_PRIO=$1 LANES=$2: $(ZL_TUNE_SMALL_LANE_PRIO=$1 ZL_TUNE_SIDE_LANES=$2 ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth | sed 's/Groth16 //' | tr '\n' ' ')"
done
done
echo "HWQ=8 SMALL_LANE_PRIO=1 LANES=4: $(GPU_MAX_HW_QUEUES=8 ZL_TUNE_SMALL_LANE_PRIO=1 ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth | sed 's/Groth16 //' | tr '\n' ' ')"
echo "HWQ=8 SMALL_LANE_PRIO=0 LANES=4: $(GPU_MAX_HW_QUEUES=8 ZL_TUNE_SMALL_LANE_PRIO=0 ITERS=40 python tools/small_lat.py g16 2>&1 | grep Groth | sed 's/Groth16 //' | tr '\n' ' ')"
for p in 1 0; do echo "SMALL_LANE_PRIO=$p msm batches: $(ZL_TUNE_SMALL_LANE_PRIO=$p BATCH=6 CS=0 python tools/msm_sweep.py 16 18 2>&1 | grep BATCH | tr '\n' ' ')"; done
