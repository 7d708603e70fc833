"""Multi-GPU MSM and NTT.

MSM: one process per GPU, each owning a contiguous shard of (bases, scalars).

Every rank runs the complete local Pippenger down to one un-normalised partial sum (ZL_PARTIAL_WORDS u64 = one XYZZ
point), the partials are all-gathered (RCCL on GPUs: ncclAllGather of raw u64 words -- elliptic-curve addition cannot
be an RCCL reduce op, so gather-then-add IS the reduce) and folded identically on every rank by zl_partials_sum.
Communication is O(100 B) per rank, so scaling is set by the local MSM alone.

NTT: one 2^log_n transform spread over G = 2^log_g ranks is the four-step factorisation with exactly one exchange
(SURVEY.md §8e): cross-rank G-point transform + twiddle (zl_ntt_cross_dev), all_to_all_single of G chunks of M/G elements
(RCCL; each rank sends (G-1)/G of its M*32 bytes over xGMI), local M-point transform (zl_ntt_dev).  Coefficients live in the
"block-column" layout, evaluations in the "cyclic" layout (include/zl_backend.h); pointwise work between transforms (the
QAP witness map) is layout-agnostic, so a chain of transforms never needs a second exchange.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import numpy as np

from .backend import ZL_G1, ZL_PARTIAL_WORDS, load_library, _p64
import ctypes as C


def forced_collective() -> bool:
    """ZL_FORCE_COLLECTIVE=1: a ONE-rank process group still goes through the collectives (all_gather_into_tensor / all_to_all_single on the tensors the
    backend of the group wants: device memory over nccl = RCCL) instead of the world == 1 shortcuts -- first contact with RCCL on a one-GPU box
    (tests/test_gpu_bench_smoke.py; VERDICT r4 missing #1)."""
    return os.environ.get("ZL_FORCE_COLLECTIVE", "") == "1"


def _collective(dist) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced_collective())


def fold_partials(curve: int, partials: np.ndarray, group: int = ZL_G1) -> Tuple[np.ndarray, int]:
    """zl_partials_sum on host memory (no GPU needed: the fold is the backend's host tail)."""
    from .backend import FQ_LIMBS

    L = load_library()
    partials = np.ascontiguousarray(partials.reshape(-1, ZL_PARTIAL_WORDS))
    out = np.zeros(2 * group * FQ_LIMBS[curve], dtype=np.uint64)
    inf = C.c_uint8(0)
    rc = L.zl_partials_sum(curve, group, _p64(partials), partials.shape[0], _p64(out), C.byref(inf))
    if rc != 0:
        raise RuntimeError(f"zl_partials_sum failed: {rc}")
    return out, inf.value


def sharded_msm(local_partial: Callable[[], np.ndarray], curve: int, group: int = ZL_G1, device=None) -> Tuple[np.ndarray, int]:
    """local_partial() -> this rank's ZL_PARTIAL_WORDS-u64 partial sum (Backend.msm_partial_dev on a GPU rank).
    Uses torch.distributed's default process group when it is initialised (nccl = RCCL on GPUs, gloo in CPU tests)."""
    import torch
    import torch.distributed as dist

    part = np.ascontiguousarray(local_partial()).reshape(-1)
    assert part.size == ZL_PARTIAL_WORDS
    if _collective(dist):
        world = dist.get_world_size()
        t = torch.from_numpy(part.view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        allp = torch.empty(world * ZL_PARTIAL_WORDS, dtype=torch.int64, device=t.device)  # flat: gloo and nccl both accept it
        dist.all_gather_into_tensor(allp, t)
        parts = allp.cpu().numpy().view(np.uint64).reshape(world, ZL_PARTIAL_WORDS)
    else:
        parts = part.reshape(1, -1)
    return fold_partials(curve, parts, group)


def sharded_msm_batch(local_partials: np.ndarray, curve: int, group: int = ZL_G1, device=None):
    """K pipelined local MSMs per rank (Backend.msm_batch_partial_dev -> (K, ZL_PARTIAL_WORDS)): ONE all_gather of K partials per rank,
    then K folds on every rank.  Returns a list of K (xy, inf)."""
    import torch
    import torch.distributed as dist

    parts = np.ascontiguousarray(local_partials).reshape(-1, ZL_PARTIAL_WORDS)
    k = parts.shape[0]
    if _collective(dist):
        world = dist.get_world_size()
        t = torch.from_numpy(parts.reshape(-1).view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        allp = torch.empty(world * k * ZL_PARTIAL_WORDS, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(allp, t)
        allparts = allp.cpu().numpy().view(np.uint64).reshape(world, k, ZL_PARTIAL_WORDS)
    else:
        allparts = parts.reshape(1, k, ZL_PARTIAL_WORDS)
    return [fold_partials(curve, np.ascontiguousarray(allparts[:, j, :]), group) for j in range(k)]


# ---- distributed NTT -------------------------------------------------------------------------------------------
def block_column_slice(x: np.ndarray, log_g: int, rank: int) -> np.ndarray:
    """This rank's part of a natural-order vector x (N, 4) in the block-column layout: local[j1*B + c] = x[j1*M + rank*B + c]."""
    G = 1 << log_g
    M = x.shape[0] // G
    B = M // G
    return np.ascontiguousarray(x.reshape(G, G, B, -1)[:, rank].reshape(M, -1))


def cyclic_slice(X: np.ndarray, log_g: int, rank: int) -> np.ndarray:
    """This rank's part of a natural-order vector X in the cyclic layout: local[k2] = X[rank + G*k2]."""
    return np.ascontiguousarray(X[rank :: 1 << log_g])


class DeviceNttEngine:
    """The two local legs on a GPU rank: tensors are int64 device tensors of shape (M, 4) holding Fr limbs."""

    def __init__(self, backend, curve: int):
        self.backend, self.curve = backend, curve

    def cross(self, t, log_n, log_g, rank, flags):
        self.backend.ntt_cross_dev(self.curve, t.data_ptr(), log_n, log_g, rank, flags)

    def local(self, t, log_m, flags):
        self.backend.ntt_dev_flags(self.curve, t.data_ptr(), log_m, flags)

    def sync(self):
        self.backend.sync()


def sharded_ntt(engine, local, log_n: int, inverse: bool = False, coset: bool = False, mont: bool = False, group=None):
    """One rank's part of a 2^log_n-point transform over the ranks of `group` (world size G = 2^log_g, 1 <= log_g <= 4).

    forward: `local` holds this rank's block-column slice of the coefficients; returns its cyclic slice of the evaluations.
    inverse: `local` holds its cyclic slice of the evaluations; returns its block-column slice of the coefficients.
    `local` is an int64 tensor (M, 4) (device tensor on GPU ranks; it is overwritten); `engine` supplies the local legs
    (DeviceNttEngine on GPUs).  `mont`: elements are Montgomery limbs on both sides."""
    import torch
    import torch.distributed as dist
    from .backend import ZL_COSET, ZL_INVERSE, ZL_MONT, ZL_MONT_IN, ZL_MONT_OUT

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    log_g = world.bit_length() - 1
    one_rank = world == 1 and forced_collective()  # G = 1: the cross step is the identity, the exchange a one-rank all_to_all_single, the local leg the whole transform
    if not one_rank and ((1 << log_g) != world or not 1 <= log_g <= 4 or 2 * log_g > log_n):
        raise ValueError("sharded_ntt needs a power-of-two world size in [2, 16] with G*G <= N")
    M = 1 << (log_n - log_g)
    if tuple(local.shape) != (M, 4):
        raise ValueError(f"local slice must have shape ({M}, 4)")
    base = (ZL_INVERSE if inverse else 0) | (ZL_COSET if coset else 0)
    plain = ZL_INVERSE if inverse else 0
    out = torch.empty_like(local)

    def exchange():
        # the legs run on the backend's stream, the collective on torch's: fence on both sides of it
        engine.sync()
        dist.all_to_all_single(out.view(-1), local.view(-1), group=group)
        if out.is_cuda:
            torch.cuda.current_stream(out.device).synchronize()

    if one_rank:
        exchange()
        engine.local(out, log_n, base | (ZL_MONT if mont else 0))
    elif not inverse:
        engine.cross(local, log_n, log_g, rank, base | (ZL_MONT if mont else ZL_MONT_OUT))
        exchange()
        engine.local(out, log_n - log_g, plain | (ZL_MONT if mont else ZL_MONT_IN))
    else:
        engine.local(local, log_n - log_g, plain | (ZL_MONT if mont else ZL_MONT_OUT))
        exchange()
        engine.cross(out, log_n, log_g, rank, base | (ZL_MONT if mont else ZL_MONT_IN))
    engine.sync()
    return out
