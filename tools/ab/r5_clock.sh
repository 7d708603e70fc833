#!/bin/bash
# round 5: effective clock of k_msm_accumulate / the multiplier chain (in-kernel s_memtime vs s_memrealtime) and of every kernel by GRBM_GUI_ACTIVE / wall
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
python -m pytest tests/test_gpu_field_kat.py tests/test_lfsr_constants_fields.py -q -m gpu 2>&1 | tail -3
python tools/clock_probe.py 24 3 > $O/r05_clock_probe_2_24.log 2>&1; tail -12 $O/r05_clock_probe_2_24.log
python tools/clock_probe.py 20 3 > $O/r05_clock_probe_2_20.log 2>&1; tail -4 $O/r05_clock_probe_2_20.log
run() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/pc_$name -o s -f csv -- "$@" > $O/pmc_clock_$name.log 2>&1
  python tools/pmc_clock.py $O/pc_$name "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -- $*" > $O/r05_pmc_clock_$name.json
  ls $O/pc_$name/* | head -5
  rm -rf $O/pc_$name
}
run msm_2_24 python tools/msm_one.py 24 0 -1 2
run ntt_2_24 python tools/ntt_one.py 24 2
python - <<'PY'
import json
for f in ("r05_pmc_clock_msm_2_24.json", "r05_pmc_clock_ntt_2_24.json"):
    d = json.load(open("gpurun_out/r5/" + f))
    print(f)
    for k, e in list(d["kernels"].items())[:8]:
        print(f"  {k[:48]:48s} {e['avg_ms']:9.3f} ms  {e['ghz_if_one_instance']:.3f} / {e['ghz_if_summed_over_8_xcds']:.3f} GHz")
PY
