"""world_size-2 gloo test of the distributed NTT driver (cross step -> all_to_all_single -> local transform) on CPU.

The two local legs are supplied by the oracle here (no GPU): the local M-point transform is oracle/zl_oracle.c's zlo_ntt and
the cross step is restated from its definition with Python integers (sums, not butterflies).  What is under test is the code
the GPU ranks run around the kernels: openzl_amd/sharded.py's layouts, flag plumbing and the exchange."""
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


class OracleNttEngine:
    """Canonical integers throughout (the Montgomery flags only choose a representation, not a value)."""

    def __init__(self, curve):
        self.curve = curve

    def cross(self, t, log_n, log_g, rank, flags):
        import oracle_lib as ol
        from oracle_lib import po
        from openzl_amd.backend import ZL_COSET, ZL_INVERSE

        r = self.curve.fr.p
        inverse, coset = bool(flags & ZL_INVERSE), bool(flags & ZL_COSET)
        G, N = 1 << log_g, 1 << log_n
        M = N // G
        B = M // G
        w = po.domain_root(self.curve, log_n)
        h = self.curve.fr_generator
        v = ol.limbs_to_ints(t.numpy().view(np.uint64))
        out = [0] * M
        for c in range(B):
            j2 = rank * B + c
            col = [v[j * B + c] for j in range(G)]
            if not inverse:
                if coset:
                    col = [col[j1] * pow(h, j1 * M + j2, r) % r for j1 in range(G)]
                for k1 in range(G):
                    acc = sum(col[j1] * pow(w, (j1 * k1 * M) % N, r) for j1 in range(G)) % r
                    out[k1 * B + c] = acc * pow(w, j2 * k1, r) % r
            else:
                winv, hinv, ginv = pow(w, r - 2, r), pow(h, r - 2, r), pow(G, r - 2, r)
                col = [col[k1] * pow(winv, j2 * k1, r) % r for k1 in range(G)]
                for j1 in range(G):
                    acc = sum(col[k1] * pow(winv, (j1 * k1 * M) % N, r) for k1 in range(G)) * ginv % r
                    if coset:
                        acc = acc * pow(hinv, j1 * M + j2, r) % r
                    out[j1 * B + c] = acc
        t.copy_(__import__("torch").from_numpy(ol.ints_to_limbs(out, 4).view(np.int64)))

    def local(self, t, log_m, flags):
        import oracle_lib as ol
        from openzl_amd.backend import ZL_INVERSE

        res = ol.oracle_ntt(self.curve, t.numpy().view(np.uint64), inverse=bool(flags & ZL_INVERSE), coset=False, mont=False)
        t.copy_(__import__("torch").from_numpy(res.view(np.int64)))

    def sync(self):
        pass


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_lib as ol
    from oracle_lib import po
    from openzl_amd.sharded import block_column_slice, cyclic_slice, sharded_ntt

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log_g = world.bit_length() - 1
    ok = True
    for curve in (po.BLS12_381, po.BN254):
        eng = OracleNttEngine(curve)
        for log_n in (2 * log_g, 6):
            x = ol.random_scalars(curve, 1 << log_n, 40 + log_n)
            for coset in (False, True):
                X = ol.oracle_ntt(curve, x, inverse=False, coset=coset)
                mine = torch.from_numpy(block_column_slice(x, log_g, rank).view(np.int64).copy())
                got = sharded_ntt(eng, mine, log_n, inverse=False, coset=coset)
                ok &= bool((got.numpy().view(np.uint64) == cyclic_slice(X, log_g, rank)).all())
                # and back: evaluations (cyclic) -> coefficients (block-column)
                back = sharded_ntt(eng, got.clone(), log_n, inverse=True, coset=coset)
                ok &= bool((back.numpy().view(np.uint64) == block_column_slice(x, log_g, rank)).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_ntt_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
