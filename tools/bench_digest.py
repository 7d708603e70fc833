#!/usr/bin/env python3
"""One line of the figures an A/B of bench.py compares.   python tools/bench_digest.py <bench.json> [label]"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("groth16") or {}
small = [(s["hashes"], round(s["prove_ms"], 3), round(s.get("two_lanes", {}).get("ms_per_proof", 0), 3)) for s in g.get("small_circuits", [])]
fk = d.get("msm_fixed_key") or {}
print((sys.argv[2] + ": ") if len(sys.argv) > 2 else "",
      f"value {d['value']:.4g} step {d['ms_per_step']:.2f} kernel {d['roofline'].get('kernel_ms', 0):.2f} single {d['config']['single_call_latency_ms']:.2f}",
      f"| ntt {(d.get('ntt') or {}).get('forward_ms', 0):.3f} | pcie {(d.get('pcie_inclusive') or {}).get('ms_per_msm', 0):.2f} | fixed {fk.get('ms_per_step', 0):.2f}",
      f"| g16 {g.get('prove_ms', 0):.2f} two {g.get('two_lanes', {}).get('ms_per_proof', 0):.2f} small {small} bn {(g.get('bn254') or {}).get('prove_ms', 0):.2f}",
      f"| c1 {((d.get('configs') or {}).get('1') or {}).get('gpu_single_call_ms', 0):.3f} c2 {((d.get('configs') or {}).get('2') or {}).get('single_call_ms', 0):.3f}", d.get("leg_errors") or "")
