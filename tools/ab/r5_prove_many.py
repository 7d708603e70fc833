"""Single proofs before the library's second prover lane exists, zl_groth16_prove_circuits (two lanes), single proofs while the idle lane is alive, and after zl_ctx_drop_lanes.
usage: r5_prove_many.py <hashes> <proofs>"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from openzl_amd import Backend, ZL_BLS12_381
from openzl_amd.backend import Circuit, Groth16Keys
K = int(sys.argv[1]); NP = int(sys.argv[2])
circ = Circuit(ZL_BLS12_381, K); be = Backend(0); keys = Groth16Keys(be, circ, seed=6)
def single():
    for _ in range(4): keys.prove(seed=7)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(NP): keys.prove(seed=7)
        best = min(best, (time.perf_counter() - t0) / NP)
    return best * 1e3
a = single()
keys.prove_many([7] * 6)
two = 1e9
for rep in range(3):
    t0 = time.perf_counter(); keys.prove_many([7] * (2 * NP)); two = min(two, (time.perf_counter() - t0) / (2 * NP) * 1e3)
b = single()
be.L.zl_ctx_drop_lanes(be._ctx)
c = single()
print(f"k={K}: single {a:.3f} ms | two lanes {two:.3f} ms per proof | single with the idle lane alive {b:.3f} | after zl_ctx_drop_lanes {c:.3f}")
