"""Multi-GPU MSM: one process per GPU, each owning a contiguous shard of (bases, scalars).

Every rank runs the complete local Pippenger down to one un-normalised partial sum (ZL_PARTIAL_WORDS u64 = one XYZZ
point), the partials are all-gathered (RCCL on GPUs: ncclAllGather of raw u64 words -- elliptic-curve addition cannot
be an RCCL reduce op, so gather-then-add IS the reduce) and folded identically on every rank by zl_partials_sum.
Communication is O(100 B) per rank, so scaling is set by the local MSM alone.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

from .backend import ZL_G1, ZL_PARTIAL_WORDS, load_library, _p64
import ctypes as C


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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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
