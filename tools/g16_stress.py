import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
torch.cuda.init()
from openzl_amd import Backend, ZL_BLS12_381, Circuit, Groth16Keys
be = Backend(0)
for k in (2300, 64):
    circ = Circuit(ZL_BLS12_381, k)
    keys = Groth16Keys(be, circ, seed=1)
    pub = circ.arrays()["assignment"][1:2]
    ref = None
    bad = 0
    for it in range(30 if k > 100 else 100):
        p, _, _ = keys.prove(seed=3)
        if ref is None:
            ref = p
            assert keys.verify(p, pub)
        elif not all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(p, ref)):
            bad += 1
    print("k", k, "mismatching proofs:", bad)
    keys.close(); circ.close()
