"""Throughput of L prover lanes (zl_ctx_fork) over ONE device-resident proving key, one host thread per lane, against one prover alone.
usage: r5_lanes.py <hashes> <proofs per lane> <max lanes>"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from openzl_amd import Backend, ZL_BLS12_381
from openzl_amd.backend import Circuit, Groth16Keys

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 12
LMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 3
circ = Circuit(ZL_BLS12_381, K)
be = Backend(0)
keys = Groth16Keys(be, circ, seed=6)
lanes = [None] + [be.fork() for _ in range(LMAX - 1)]
ref = keys.prove(seed=7)[0]
for ln in lanes:
    for _ in range(3):
        p = keys.prove(seed=7, lane=ln)[0]
        assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(ref, p))

def run(ln, n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        p = keys.prove(seed=7, lane=ln)[0]
    out.append((time.perf_counter() - t0, p))

alone = None
for L in range(1, LMAX + 1):
    best = None
    for rep in range(3):
        outs = [[] for _ in range(L)]
        th = [threading.Thread(target=run, args=(lanes[i], NP, outs[i])) for i in range(L)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        wall = (time.perf_counter() - t0) / (L * NP)
        for oo in outs:
            assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(ref, oo[0][1]))
        best = wall if best is None else min(best, wall)
    alone = alone or best
    print(f"k={K} lanes={L}: {best * 1e3:.3f} ms per proof (throughput, best of 3)  x{alone / best:.3f}")
for ln in lanes[1:]:
    ln.close()
keys.close(); be.close()
