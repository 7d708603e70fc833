#!/usr/bin/env python3
"""A few Groth16 proofs of the config-5 circuit, for a rocprofv3 kernel trace (tools/g16_timeline.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

torch.cuda.init()
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254, Circuit, Groth16Keys

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
be = Backend(0)
be.enable_timing(True)
if os.environ.get("PRE_LEGS"):
    from pre_legs import run_pre_legs
    run_pre_legs(be)
t0 = time.perf_counter(); circ = Circuit(ZL_BN254 if os.environ.get("CURVE") == "bn254" else ZL_BLS12_381, k); t1 = time.perf_counter()
keys = Groth16Keys(be, circ, seed=1); t2 = time.perf_counter()
print(f"synthesis {t1 - t0:.2f} s  setup {t2 - t1:.2f} s", flush=True)
ts = []
for _ in range(int(os.environ.get("ITERS", "4"))):
    t0 = time.perf_counter()
    keys.prove(seed=3)
    ts.append((time.perf_counter() - t0) * 1e3)
    print(f"prove {ts[-1]:.2f} ms (device {be.last_timing().total_ms:.2f})", flush=True)
print(f"prove k={k}: min {min(ts[1:] or ts):.2f}  median {sorted(ts[1:] or ts)[len(ts[1:] or ts) // 2]:.2f} ms over {len(ts[1:] or ts)} (first dropped)", flush=True)
if os.environ.get("G16_WIRE"):
    # ProvingContext wire format at full size: encode (queries downloaded from the device), decode (upload + window tables), prove
    from openzl_amd import backend as zb
    t0 = time.perf_counter(); data = keys.to_bytes(); t1 = time.perf_counter()
    p1, _, _ = keys.prove(seed=5)
    keys.close()
    t2 = time.perf_counter(); keys2 = Groth16Keys.from_bytes(be, circ, data); t3 = time.perf_counter()
    p2, _, _ = keys2.prove(seed=5)
    same = zb.proof_to_bytes(ZL_BLS12_381, p1) == zb.proof_to_bytes(ZL_BLS12_381, p2)
    for _ in range(3):
        t4 = time.perf_counter(); keys2.prove(seed=3); t5 = time.perf_counter()
    print(f"wire: {len(data) / 1e6:.1f} MB  encode {t1 - t0:.2f} s  decode {t3 - t2:.2f} s  prove after decode {(t5 - t4) * 1e3:.2f} ms  same proof: {same}", flush=True)
