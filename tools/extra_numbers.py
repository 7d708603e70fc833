#!/usr/bin/env python3
"""Secondary numbers quoted in DESIGN.md: PCIe-inclusive MSM (host scalars), BN254 MSM / NTT throughput."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381, ZL_BN254

R_BN = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
be = Backend(0); be.enable_timing(True)
dev = torch.device("cuda", 0)
ln = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << ln
for name, cid, r, bits in (("bls12_381", ZL_BLS12_381, None, 255), ("bn254", ZL_BN254, R_BN, 254)):
    kw = {} if r is None else {"r": r, "bits": bits}
    k = random_scalars_lt_r(n, 1, **kw); s = random_scalars_lt_r(n, 2, **kw)
    h = be.bases_generate(cid, k)
    ds = torch.from_numpy(s.view(np.int64)).to(dev)
    be.msm_dev(h, ds.data_ptr(), n)
    t0 = time.perf_counter(); be.msm_dev(h, ds.data_ptr(), n); t_dev = time.perf_counter() - t0
    be.msm(h, s)
    t0 = time.perf_counter(); be.msm(h, s); t_host = time.perf_counter() - t0
    be.bases_precompute(h, 0)
    be.msm_dev(h, ds.data_ptr(), n)
    t0 = time.perf_counter(); be.msm_dev(h, ds.data_ptr(), n); t_pre = time.perf_counter() - t0
    x = torch.from_numpy(random_scalars_lt_r(n, 3, **kw).view(np.int64)).to(dev)
    be.ntt_dev(cid, x.data_ptr(), ln); be.ntt_dev(cid, x.data_ptr(), ln)
    t_ntt = be.last_timing().total_ms
    print(f"{name} 2^{ln}: MSM resident {t_dev*1e3:.2f} ms ({n/t_dev/1e6:.0f} Mpts/s), host scalars (PCIe-inclusive) {t_host*1e3:.2f} ms, "
          f"with table {t_pre*1e3:.2f} ms ({n/t_pre/1e6:.0f} Mpts/s); NTT {t_ntt:.3f} ms ({n/(t_ntt*1e-3)/1e9:.2f} Gel/s)", flush=True)
    be.bases_free(h)
