"""Multi-GPU: replicas are independent, so the batch is block-partitioned over ranks (one process per GPU) and the ONLY
communication is one all-gather of the per-replica results after the kernel: Morris ``Y[B]`` or
residual / objective vectors -- 8..24 bytes per replica.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI);
"gloo" is used by the CPU tests of this module's partition / gather logic.

The reference's counterpart is the ProcessPoolExecutor fan-out + result collection in sensitivity/analysis.py:241-259."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``total`` replicas owned by ``rank``: ceil-sized blocks, last ranks may be short / empty."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    per = -(-total // world)
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def all_gather_replicas(local: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Gather each rank's per-replica results (first dim = its shard, in ``shard_bounds`` order) into the full [total, ...]
    tensor on every rank with ONE collective.  Short shards are padded to the common block size so that a single
    ``all_gather_into_tensor`` (ncclAllGather) suffices."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != total:
            raise ValueError("single-rank gather: local shard must be the whole batch")
        return local
    world = dist.get_world_size(group)
    per = -(-total // world)
    tail = local.shape[1:]
    if local.shape[0] != per:
        pad = torch.zeros((per - local.shape[0],) + tuple(tail), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((world * per,) + tuple(tail), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:total]


def sharded_map(fn: Callable[[int, int], torch.Tensor], total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Run ``fn(lo, hi)`` on this rank's shard and all-gather the per-replica results.  ``fn`` returns [hi - lo, ...]."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(total, rank, world)
    return all_gather_replicas(fn(lo, hi), total, group)
