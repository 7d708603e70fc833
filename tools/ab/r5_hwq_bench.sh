for q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu --no-live-traffic > gpurun_out/bench_hwq$q.json 2> gpurun_out/bench_hwq$q.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_hwq$q.json").read().strip().splitlines()[-1])
g=d["groth16"]
print("HWQ=$q value %.4g ms %.2f kernel %.2f | ntt %.3f | pcie %.2f | fixed %.2f | g16 %.2f two %.2f | small %s | bn %.2f | c1 %.3f/%.3f c2 %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["ntt"]["forward_ms"], d["pcie_inclusive"]["ms_per_msm"], d["msm_fixed_key"]["ms_per_step"], g["prove_ms"], g["two_lanes"]["ms_per_proof"], [(s["hashes"], round(s["prove_ms"],3), round(s["two_lanes"]["ms_per_proof"],3)) for s in g["small_circuits"]], g["bn254"]["prove_ms"], d["configs"]["1"]["gpu_single_call_ms"], d["configs"]["1"].get("gpu_pipelined_ms_per_msm",0), d["configs"]["2"]["single_call_ms"]), d.get("leg_errors"))
PY
done
