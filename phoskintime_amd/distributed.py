"""Multi-GPU: replicas are independent, so the batch is partitioned over ranks (one process per GPU) and the ONLY
communication is one all-gather of the per-replica results after the kernel: Morris ``Y[B]`` or
residual / objective vectors -- 8..24 bytes per replica.  Two partitions: contiguous blocks (``shard_bounds`` -- bench.py's weak-scaling
batches, where every rank draws its own replicas), and INTERLEAVED rows (``interleaved_rows``: rank r owns rows r, r + W, r + 2 W ... of
an optional cost order) for the drivers whose rows differ in cost -- a Morris design walks through parameter space trajectory by
trajectory, a population carries its stiff candidates wherever the optimiser put them, and the workgroup-per-replica kernels take
30 .. 8 000 steps per row: with blocks one rank can own the stragglers, with interleaving every rank gets the same mix.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI);
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


def interleaved_rows(total: int, rank: int, world: int, order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Global row indices owned by ``rank`` under the interleaved partition: positions rank, rank + world, ... of ``order`` (a permutation
    of range(total), e.g. ``cost_order``; identity when None).  int64, on ``order``'s device (CPU when None)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    pos = torch.arange(rank, max(total, rank), world, dtype=torch.int64, device=None if order is None else order.device)      # empty when rank >= total
    return pos if order is None else order[pos]


def cost_order(cost: torch.Tensor) -> torch.Tensor:
    """Rows by decreasing cost proxy (stable): dealing this order round-robin gives every rank the same mix of expensive and cheap rows.
    Every rank must compute it from the SAME ``cost`` tensor (it is a pure function of the inputs all ranks hold)."""
    return torch.argsort(cost.reshape(-1), descending=True, stable=True)


def all_gather_interleaved(local: torch.Tensor, total: int, order: Optional[torch.Tensor] = None,
                           group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Inverse of ``interleaved_rows`` with ONE collective: every rank passes the results of its rows (in its local order) and receives the
    full [total, ...] tensor in GLOBAL row order.  Rank r's j-th row sits at position j * world + r of ``order``."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if local.shape[0] != total:
            raise ValueError("single-rank gather: local rows must be the whole batch")
        if order is None:
            return local
        out = torch.empty_like(local)
        out[order.to(local.device)] = local
        return out
    world = dist.get_world_size(group)
    per = -(-total // world)
    tail = tuple(local.shape[1:])
    if local.shape[0] != per:
        local = torch.cat([local, torch.zeros((per - local.shape[0],) + tail, dtype=local.dtype, device=local.device)], dim=0)
    buf = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, local.contiguous(), group=group)
    # buf[r * per + j] belongs to position j * world + r: transpose the (rank, slot) grid; positions >= total are exactly the padding
    by_pos = buf.reshape((world, per) + tail).transpose(0, 1).reshape((world * per,) + tail)[:total]
    if order is None:
        return by_pos.contiguous()
    out = torch.empty_like(by_pos)
    out[order.to(by_pos.device)] = by_pos
    return out


def all_gather_interleaved_with_status(values: torch.Tensor, status: torch.Tensor, total: int, order: Optional[torch.Tensor] = None,
                                       group: Optional[dist.ProcessGroup] = None):
    """``all_gather_interleaved`` for float64 results [n, ...] and their int32 status flags [n] through ONE collective (the flags ride as an
    extra float64 column: exact, they are small integers)."""
    width = 1
    for d in values.shape[1:]:
        width *= int(d)
    packed = torch.cat([values.reshape(values.shape[0], width).to(torch.float64), status.to(torch.float64).reshape(-1, 1)], dim=1)
    full = all_gather_interleaved(packed, total, order, group)
    return full[:, :-1].reshape((total,) + tuple(values.shape[1:])), full[:, -1].to(torch.int32)


def _world(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _control_device(group=None) -> torch.device:
    """Device a control-plane tensor must live on for the group's backend (RCCL moves only HBM tensors)."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def shared_seed(seed: Optional[int], group: Optional[dist.ProcessGroup] = None) -> Optional[int]:
    """A seed every rank agrees on.  With ``seed=None`` each rank's ``default_rng(None)`` would draw its OWN Morris design, and
    the gathered outputs would mix rows of unrelated designs: rank 0 draws 63 bits of OS entropy and broadcasts them (8 bytes, control
    plane, before the kernel -- the data path keeps its single all-gather).  Single rank or an explicit seed: returned unchanged."""
    rank, world = _world(group)
    if seed is not None or world == 1:
        return seed
    import numpy as np
    box = torch.zeros(1, dtype=torch.int64, device=_control_device(group))
    if rank == 0:
        box[0] = int(np.random.SeedSequence().entropy % (1 << 62))
    dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(box.item())


def all_gather_with_status(values: torch.Tensor, status: torch.Tensor, total: int, group: Optional[dist.ProcessGroup] = None):
    """Per-replica float64 results [n, ...] and their int32 status flags [n] through ONE all-gather: the flags ride as one extra
    float64 column (exact: they are small integers).  Returns (values [total, ...], status [total] int32)."""
    width = 1
    for d in values.shape[1:]:
        width *= int(d)
    v2 = values.reshape(values.shape[0], width).to(torch.float64)          # (an empty shard cannot be reshaped with -1)
    packed = torch.cat([v2, status.to(torch.float64).reshape(-1, 1)], dim=1)
    full = all_gather_replicas(packed, total, group)
    return full[:, :-1].reshape((total,) + tuple(values.shape[1:])), full[:, -1].to(torch.int32)


def sharded_map(fn: Callable[[int, int], torch.Tensor], total: int, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Run ``fn(lo, hi)`` on this rank's shard and all-gather the per-replica results.  ``fn`` returns [hi - lo, ...]."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(total, rank, world)
    return all_gather_replicas(fn(lo, hi), total, group)


def sharded_map_rows(fn: Callable[[torch.Tensor], torch.Tensor], total: int, cost: Optional[torch.Tensor] = None,
                     group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Interleaved counterpart of ``sharded_map``: ``fn(rows)`` computes the results [len(rows), ...] of the global rows ``rows`` (int64
    tensor); rows are dealt round-robin over the ranks, in decreasing ``cost`` when a proxy is given; ONE all-gather returns the full
    tensor in global row order on every rank -- equal to ``fn(arange(total))`` row for row when ``fn`` treats rows independently."""
    rank, world = _world(group)
    order = None if cost is None else cost_order(cost)
    rows = interleaved_rows(total, rank, world, order)
    return all_gather_interleaved(fn(rows), total, order, group)
