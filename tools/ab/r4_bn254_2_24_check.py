import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
torch.cuda.init()
import oracle_lib as ol
from oracle_lib import po
from openzl_amd import Backend
from test_gpu_msm import _dot_mod_r_u64k, _oracle_point
be = Backend(0)
curve = po.BN254
n, r = 1 << 24, curve.fr.p
rng = np.random.Generator(np.random.PCG64(77))
k64 = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
k = np.zeros((n, 4), dtype=np.uint64); k[:, 0] = k64
S = ol.random_scalars(curve, n, 78)
h = be.bases_generate(curve.cid, k)
d = torch.from_numpy(S.view(np.int64)).cuda(); torch.cuda.synchronize()
got, inf = be.msm_dev(h, d.data_ptr(), n)
parts = be.msm_batch_partial_dev(h, [d.data_ptr()] * 3, n)
exp = _oracle_point(curve, _dot_mod_r_u64k(S, k64, r))
ok = (not inf) and (got == exp).all()
for j in range(3):
    xy, pinf = be.partials_sum(curve.cid, parts[j:j + 1]); ok = ok and (not pinf) and (xy == exp).all()
print("BN254 2^24 known-dlog exact:", bool(ok))
