"""Repeat proofs of several circuit sizes over both curves and demand byte-identical proofs (fixed r, s) that verify: a race between the streams / host threads of a
proof (phase-major issue, lanes sharing hardware queues, persistent workers) shows up as a mismatch.  Interleaves sizes so that buffers are re-planned between proofs.
    python tools/g16_stress.py [rounds=3]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

torch.cuda.init()
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254, Circuit, Groth16Keys

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
be = Backend(0)
cases = []
for curve, name in ((ZL_BLS12_381, "bls12_381"), (ZL_BN254, "bn254")):
    for k in (1, 64, 600, 4096):
        circ = Circuit(curve, k)
        keys = Groth16Keys(be, circ, seed=1)
        pub = circ.arrays()["assignment"][1:2]
        ref, _, _ = keys.prove(seed=3)
        assert keys.verify(ref, pub), (name, k)
        cases.append((name, k, circ, keys, ref))
        print("ready", name, k, flush=True)
bad = 0
total = 0
for rnd in range(rounds):
    for name, k, circ, keys, ref in cases:
        for it in range(4 if k > 1000 else 12):
            p, _, _ = keys.prove(seed=3)
            total += 1
            if not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p, ref)):
                bad += 1
                print("MISMATCH", name, k, rnd, it, flush=True)
print(f"{total} proofs, mismatching: {bad}")
sys.exit(1 if bad else 0)
