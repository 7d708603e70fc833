#!/usr/bin/env python3
"""Groth16 prove of the 4096-hash Poseidon circuit with and without precomputed window tables on the proving key's five query vectors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from openzl_amd import Backend, ZL_BLS12_381, Circuit, Groth16Keys
be = Backend(0); be.enable_timing(True)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else []
circ = Circuit(ZL_BLS12_381, k); keys = Groth16Keys(be, circ, seed=1)
def timeit(tag):
    p0, _, _ = keys.prove(seed=3)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); p, _, _ = keys.prove(seed=3); ts.append(time.perf_counter() - t0)
    print(f"{tag}: prove min {min(ts)*1e3:.2f} med {np.median(ts)*1e3:.2f} ms", flush=True)
    return p
p_plain = timeit("plain")  # "plain" = the keys as Groth16::compile leaves them (window tables for keys above 2^19 points: ZL_TUNE_G1_TABLE_C / ZL_TUNE_G2_TABLE_C)
for c in cs:
    t0 = time.perf_counter()
    for name in ("a_query", "b_g1_query", "h_query", "l_query", "b_g2_query"):
        be.bases_precompute(getattr(keys.pk, name), c)
    torch.cuda.synchronize()
    print(f"tables c={c} built in {(time.perf_counter()-t0)*1e3:.1f} ms; free mem {torch.cuda.mem_get_info()[0]/2**30:.1f} GiB")
    p_tab = timeit(f"tables c={c}")
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(p_plain, p_tab)), "proof differs"
print("proofs identical")
