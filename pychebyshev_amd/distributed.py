"""Sharding of query batches across the GPUs of one node (one process per GPU).

The path has no exchange step: every query point is independent and the model (<= 1.3 MB)
is replicated, so rank g evaluates a contiguous row block of ``points`` on its own GPU.
The only collective is the final gather of the per-rank result blocks
(``torch.distributed`` -- RCCL over xGMI with the ``nccl`` backend, ``gloo`` on CPU for
tests).  ``torch`` is imported lazily and only here: it is launch/collective plumbing.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_bounds(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of rank ``rank``: blocks of ``ceil(N / G)`` rows."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    per = -(-n_rows // world_size)
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def gather_results(local: "np.ndarray", n_rows: int, group=None, dst: int = 0):
    """Gather per-rank result blocks (layout of :func:`shard_bounds`) on rank ``dst``.

    ``local`` may be a NumPy array (CPU / gloo) or a torch tensor already on this rank's
    GPU (nccl = RCCL).  Returns the full ``(n_rows,)`` array on ``dst`` and ``None`` elsewhere.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-n_rows // world)
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local
    if dist.get_backend(group) == "nccl" and t.device.type == "cpu":
        t = t.cuda()                        # RCCL moves device memory: stage the host block on this rank's GPU
    buf = torch.zeros(per, dtype=t.dtype, device=t.device)   # equal-size blocks for gather
    buf[: t.numel()] = t
    if dist.get_backend(group) == "nccl":
        out = torch.empty(per * world, dtype=t.dtype, device=t.device) if rank == dst else None
        outs = list(out.split(per)) if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
    else:
        outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, outs, dst=dst, group=group)
        out = torch.cat(outs) if rank == dst else None
    if rank != dst:
        return None
    full = out[:n_rows]
    return full.cpu().numpy() if is_np else full


def eval_sharded(evaluate: Callable[["np.ndarray"], "np.ndarray"], points: "np.ndarray", group=None,
                 dst: int = 0):
    """Evaluate ``points`` (same array on every rank) block-wise: each rank runs
    ``evaluate`` on its own row block; results are gathered on ``dst``."""
    import torch.distributed as dist

    lo, hi = shard_bounds(points.shape[0], dist.get_rank(group), dist.get_world_size(group))
    local = np.asarray(evaluate(points[lo:hi]), dtype=np.float64)
    return gather_results(local, points.shape[0], group=group, dst=dst)
