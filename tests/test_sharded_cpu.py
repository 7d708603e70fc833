"""world_size-2 gloo test of the N>1 MSM path (shard -> local partial -> all_gather -> fold) on CPU.  The local
partial is produced by the oracle here (no GPU); the gather + zl_partials_sum fold is the code the GPU ranks run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import numpy as np
    import torch.distributed as dist

    import oracle_lib as ol
    from oracle_lib import po
    from openzl_amd.sharded import sharded_msm
    from test_abi import _partial_from_affine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    curve = po.BLS12_381
    n = 256  # total; contiguous shards of n/world
    K = ol.random_scalars(curve, n, 5)
    S = ol.random_scalars(curve, n, 6)
    B = ol.oracle_g1_mul_gen(curve, K)
    lo, hi = rank * n // world, (rank + 1) * n // world

    def local_partial():
        xy, inf = ol.oracle_msm_g1(curve, B[lo:hi], S[lo:hi], 0, 1)
        return _partial_from_affine(curve, ol.limbs_to_point(curve, xy, inf))

    xy, inf = sharded_msm(local_partial, curve.cid)
    full, finf = ol.oracle_msm_g1(curve, B, S, 0, 1)
    ok = bool((xy == full).all() and inf == finf)
    # K steps per rank folded after ONE gather (what bench.py's pipelined batch does): step j uses the scalars rotated by j
    from openzl_amd.sharded import sharded_msm_batch

    K = 3
    parts, fulls = [], []
    for j in range(K):
        Sj = np.roll(S, j, axis=0)
        pxy, pinf = ol.oracle_msm_g1(curve, B[lo:hi], Sj[lo:hi], 0, 1)
        parts.append(_partial_from_affine(curve, ol.limbs_to_point(curve, pxy, pinf)))
        fulls.append(ol.oracle_msm_g1(curve, B, Sj, 0, 1))
    res = sharded_msm_batch(np.stack(parts), curve.cid)
    for (gxy, ginf), (fxy, finf2) in zip(res, fulls):
        ok = ok and bool((gxy == fxy).all() and ginf == finf2)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]
