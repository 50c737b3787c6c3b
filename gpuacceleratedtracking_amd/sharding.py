"""Multi-GPU: satellite channels shard embarrassingly across devices.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).
Every rank holds the full antenna signal (replicated) and correlates only ITS contiguous slice of
the K channels; outputs are disjoint, so the data path needs NO collective.  The only
communication is control-plane: timing barriers in bench.py and an optional ``gather_outputs``
(all_gather of the tiny [B, K_r, L, M] results) for a host that wants them in one place.

The reference itself is single-device (grep finds no multi-device call site; the closest is the
multi-satellite kernel ``downconvert_and_correlate_kernel_3d_4431!``, src/algorithms.jl:637).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ShardPlan:
    """Contiguous split of ``total`` units (channels, or time blocks when K < world) over ranks."""

    total: int
    world_size: int
    rank: int

    def __post_init__(self):
        if self.total < 0 or self.world_size < 1 or not 0 <= self.rank < self.world_size:
            raise ValueError("bad shard plan")

    def bounds(self, rank: int | None = None) -> tuple[int, int]:
        r = self.rank if rank is None else rank
        base, extra = divmod(self.total, self.world_size)
        lo = r * base + min(r, extra)
        return lo, lo + base + (1 if r < extra else 0)

    @property
    def lo(self) -> int:
        return self.bounds()[0]

    @property
    def hi(self) -> int:
        return self.bounds()[1]

    @property
    def count(self) -> int:
        lo, hi = self.bounds()
        return hi - lo

    def counts(self) -> list[int]:
        return [self.bounds(r)[1] - self.bounds(r)[0] for r in range(self.world_size)]


def shard_channels(num_channels: int, world_size: int, rank: int) -> ShardPlan:
    return ShardPlan(num_channels, world_size, rank)


def shard_params(params: np.ndarray, plan: ShardPlan) -> np.ndarray:
    """Slice a [B, K] parameter array down to this rank's channels -> [B, K_r]."""
    if params.ndim != 2 or params.shape[1] != plan.total:
        raise ValueError("params must be [B, K] with K == plan.total")
    return np.ascontiguousarray(params[:, plan.lo:plan.hi])


def gather_outputs(local: np.ndarray, plan: ShardPlan, group=None) -> np.ndarray:
    """Concatenate per-rank outputs [B, K_r, L, M] along the channel axis on every rank.
    Control-plane convenience only (a few KB); uses all_gather_object so ragged K_r is fine and
    it works on any backend."""
    import torch.distributed as dist

    if not dist.is_initialized() or plan.world_size == 1:
        return local
    parts = [None] * plan.world_size
    dist.all_gather_object(parts, local, group=group)
    return np.concatenate([p for p in parts if p.shape[1] > 0], axis=1)


def max_over_ranks(value: float, device=None, group=None) -> float:
    """MAX-reduce a scalar (timing) across ranks."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
