"""What bench.py does in front of its Groth16 leg: a pipelined batch (sort / tail streams) and an MSM with host scalars (copy stream).  The streams they
create take part in the runtime's stream -> hardware-queue assignment of everything created later (PRE_LEGS=1 in tools/g16_one.py, tools/g16_lat_dist.py)."""
import numpy as np
import torch


def run_pre_legs(be, log_n: int = 22):
    from bench import random_scalars_lt_r
    from openzl_amd import ZL_BLS12_381

    n = 1 << log_n
    rng = np.random.Generator(np.random.PCG64(1))
    kk = np.zeros((n, 4), dtype=np.uint64)
    kk[:, 0] = rng.integers(1, 1 << 63, size=n, dtype=np.uint64)
    h = be.bases_generate(ZL_BLS12_381, kk)
    sc = random_scalars_lt_r(n, 2)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    be.msm_batch_partial_dev(h, [d.data_ptr()] * 3, n)
    be.msm(h, sc)
    be.bases_free(h)
    del d
