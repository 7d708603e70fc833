"""Single-proof latency in a process that loads only the library (no torch).  usage: r5_single_notorch.py <hashes> <proofs>"""
import sys, time
sys.path.insert(0, ".")
from openzl_amd import Backend, ZL_BLS12_381
from openzl_amd.backend import Circuit, Groth16Keys
K = int(sys.argv[1]); NP = int(sys.argv[2])
circ = Circuit(ZL_BLS12_381, K); be = Backend(0); keys = Groth16Keys(be, circ, seed=6)
for _ in range(4): keys.prove(seed=7)
ts = []
for _ in range(NP):
    t0 = time.perf_counter(); keys.prove(seed=7); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"k={K}: min {ts[0] * 1e3:.3f} median {ts[len(ts) // 2] * 1e3:.3f} ms")
