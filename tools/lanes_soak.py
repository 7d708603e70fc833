#!/usr/bin/env python3
"""Soak of the prover lanes: T host threads (the ctx and T - 1 forks) prove a mix of circuit sizes and witnesses over shared device-resident keys for a few minutes;
every proof must equal the reference proof of its (circuit, witness, seed) made beforehand on the parent alone.   python tools/lanes_soak.py [seconds = 60] [threads = 3]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254
from openzl_amd.backend import Circuit, Groth16Keys

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
be = Backend(0)
lanes = [None] + [be.fork() for _ in range(T - 1)]
cases = []
for curve, k in ((ZL_BLS12_381, 1), (ZL_BLS12_381, 64), (ZL_BN254, 8), (ZL_BLS12_381, 512)):
    full = Circuit(curve, k)
    keys = Groth16Keys(be, full, seed=1000 + k)
    wits = [Circuit(curve, k, x0=3 + j, x1=5 * j, witness_only=True) for j in range(3)]
    for j, w in enumerate([full] + wits):
        for seed in (11, 12):
            cases.append((keys, w, seed, keys.prove(seed=seed, circuit=w)[0]))
print(f"{len(cases)} (key, witness, seed) cases, {T} threads, {secs:.0f} s", flush=True)
stop = time.time() + secs
bad, count, errs = [], [0] * T, []
def same(p, q):
    return all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p, q))
def run(t):
    rng = np.random.Generator(np.random.PCG64(t))
    try:
        while time.time() < stop:
            keys, w, seed, ref = cases[int(rng.integers(0, len(cases)))]
            if not same(ref, keys.prove(seed=seed, circuit=w, lane=lanes[t])[0]):
                bad.append((t, seed))
            count[t] += 1
    except Exception as e:  # noqa: BLE001
        errs.append(repr(e))
th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
[t.start() for t in th]
[t.join() for t in th]
print(f"proofs per thread {count}, mismatches {len(bad)}, errors {errs}")
sys.exit(1 if bad or errs else 0)
