"""Experiment (round 5): do two pipelined MSM batches on two lanes (zl_ctx_fork) of one device finish sooner than one batch of twice the length?
usage: r5_msm_two_lanes.py <log_n> <steps per lane>"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import torch
from bench import random_scalars_lt_r
from openzl_amd import Backend, ZL_BLS12_381

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << log_n
be = Backend(0)
lane = be.fork()
k = random_scalars_lt_r(n, 1)
h = be.bases_generate(ZL_BLS12_381, k)
d = [torch.from_numpy(random_scalars_lt_r(n, 2 + j).view(np.int64)).cuda() for j in range(2)]
torch.cuda.synchronize()
ptrs = [d[i % 2].data_ptr() for i in range(2 * K)]
ref = be.msm_batch_partial_dev(h, ptrs[:4], n)
lane.msm_batch_partial_dev(h, ptrs[:4], n)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    one = be.msm_batch_partial_dev(h, ptrs, n)
    t_one = time.perf_counter() - t0
    outs = [None, None]
    def run(b, i):
        outs[i] = b.msm_batch_partial_dev(h, ptrs[:K], n)
    th = [threading.Thread(target=run, args=(b, i)) for i, b in enumerate((be, lane))]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    t_two = time.perf_counter() - t0
    aff = lambda parts, j: be.partials_sum(ZL_BLS12_381, parts[j:j + 1])[0]  # (partials are un-normalised: compare the affine points)
    ok = all((aff(o, j) == aff(ref, j)).all() for o in outs + [one] for j in range(4))
    print(f"2^{log_n}: one lane x {2 * K} steps {t_one / (2 * K) * 1e3:.3f} ms/step   two lanes x {K} steps {t_two / (2 * K) * 1e3:.3f} ms/step   exact: {ok}")
lane.close(); be.bases_free(h)
