#!/bin/bash
# SQ issue / stall counters of every kernel of one plain 2^24 MSM and of one 2^16 MSM (own pass; --pmc with --kernel-trace only)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/psq -o s -f csv -- python tools/msm_one.py 24 0 -1 1 > $O/pmc_sq_24.log 2>&1
python tools/pmc_sq.py $(find $O/psq -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $C --kernel-trace -f csv -- python tools/msm_one.py 24 0 -1 1" > $O/r03_pmc_sq_msm_2_24.json
timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/psq2 -o s -f csv -- python tools/msm_one.py 16 0 -1 2 > $O/pmc_sq_16.log 2>&1
python tools/pmc_sq.py $(find $O/psq2 -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $C --kernel-trace -f csv -- python tools/msm_one.py 16 0 -1 2" > $O/r03_pmc_sq_msm_2_16.json
rm -rf $O/psq $O/psq2
python - <<'PY'
import json
for f in ("r03_pmc_sq_msm_2_24.json", "r03_pmc_sq_msm_2_16.json"):
    d = json.load(open("gpurun_out/final/" + f))
    print(f)
    for k, e in list(d["kernels"].items())[:8]:
        print(f"  {k[:40]:40s} issuing {e['frac_issuing']:.2f} issue-stalled {e['frac_issue_stalled']:.2f} parked {e['frac_parked_on_waitcnt']:.2f} valu/issuing {e['frac_valu_of_issuing']:.2f}")
PY
