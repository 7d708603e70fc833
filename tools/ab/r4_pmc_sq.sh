#!/bin/bash
# round 4: SQ issue / stall counters (own --pmc pass, --kernel-trace only) of the 2^24 BLS12-381 MSM, of the 2^24 lazy NTT and of the G2 MSM 2^20
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES"
run() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/psq_$name -o s -f csv -- "$@" > $O/pmc_sq_$name.log 2>&1
  python tools/pmc_sq.py $(find $O/psq_$name -name "*counter_collection.csv" | head -1) "rocprofv3 --pmc $C --kernel-trace -f csv -- $*" > $O/r04_pmc_sq_$name.json
  rm -rf $O/psq_$name
}
run msm_2_24 python tools/msm_one.py 24 0 -1 1
run ntt_2_24 python tools/ntt_one.py 24 2
run g2_2_20 python tools/msm_sweep.py --g2 20
python - <<'PY'
import json
for f in ("r04_pmc_sq_msm_2_24.json", "r04_pmc_sq_ntt_2_24.json", "r04_pmc_sq_g2_2_20.json"):
    d = json.load(open("gpurun_out/final/" + f))
    print(f)
    for k, e in list(d["kernels"].items())[:5]:
        print(f"  {k[:44]:44s} issuing {e['frac_issuing']:.2f} issue-stalled {e['frac_issue_stalled']:.2f} parked {e['frac_parked_on_waitcnt']:.2f} valu/issuing {e['frac_valu_of_issuing']:.2f}")
PY
