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


class DeviceGroup:
    """Several devices driven from ONE host thread through the C ABI's group entry points (include/gat.h
    ``gat_group_*``): the in-process counterpart of the one-process-per-GPU path above, and what a C / Julia host
    calls (examples/gat_multi_gpu.c).  Member r correlates the contiguous channel slice ``shard(K, r)`` on ITS copy
    of the signal; the signal is replicated with peer copies (``replicate``), outputs are disjoint -- no collective.
    ``devices`` may name one device several times (two members on device 0 rehearse the path on a one-GPU box)."""

    def __init__(self, devices):
        import ctypes as C

        from . import _lib

        self._C, self._lib_mod = C, _lib
        self.lib = _lib.load()
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        rc = self.lib.gat_group_create(len(devices), devs, C.byref(self._h))
        if rc != 0:
            raise _lib.GatError(rc, "gat_group_create")
        self.devices = [int(d) for d in devices]

    def _check(self, rc, where):
        if rc != 0:
            msg = self.lib.gat_group_last_error(self._h)
            raise self._lib_mod.GatError(rc, where, msg.decode() if msg else "")

    def close(self):
        if self._h:
            self.lib.gat_group_destroy(self._h)
            self._h = self._C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self) -> int:
        n = self._C.c_int32()
        self._check(self.lib.gat_group_size(self._h, self._C.byref(n)), "gat_group_size")
        return n.value

    def shard(self, num_channels: int, rank: int) -> tuple[int, int]:
        """(first, count) of member ``rank`` -- the same contiguous split as ``ShardPlan.bounds``."""
        lo, cnt = self._C.c_int32(), self._C.c_int32()
        self._check(self.lib.gat_group_shard(self._h, num_channels, rank, self._C.byref(lo), self._C.byref(cnt)),
                    "gat_group_shard")
        return lo.value, cnt.value

    def set_codes(self, codes: np.ndarray):
        arr = np.ascontiguousarray(codes, dtype=np.int8)
        p, lc = arr.shape
        self._check(self.lib.gat_group_set_codes(self._h, arr.ctypes.data_as(self._C.POINTER(self._C.c_int8)), lc, p),
                    "gat_group_set_codes")

    def _ptrs(self, tensors):
        return (self._C.c_void_p * len(tensors))(*[int(t.data_ptr()) for t in tensors])

    def replicate(self, src_rank: int, tensors):
        """tensors[r] (r != src_rank) <- tensors[src_rank]; one equally sized buffer per member, on its device."""
        nbytes = tensors[src_rank].numel() * tensors[src_rank].element_size()
        if any(t.numel() * t.element_size() != nbytes for t in tensors):
            raise ValueError("replicate: buffers differ in size")
        self._check(self.lib.gat_group_replicate(self._h, src_rank, self._ptrs(tensors), nbytes), "gat_group_replicate")

    def correlate(self, descs, params: np.ndarray, shifts, sampling_frequency: float, outs_re, outs_im, flags: int = 0):
        """params: structured [B, K] for ALL channels; descs[r] / outs_*[r]: member r's signal descriptor and
        float32 output tensors ([B, K_r, L, M]) on its device.  Asynchronous on every member's stream."""
        C = self._C
        prm = np.ascontiguousarray(params, dtype=self._lib_mod.PARAMS_DTYPE)
        B, K = prm.shape
        sh = np.ascontiguousarray(shifts, dtype=np.int32)
        d = (self._lib_mod.SignalDesc * len(descs))(*descs)
        rc = self.lib.gat_group_correlate(self._h, d, prm.ctypes.data_as(C.POINTER(self._lib_mod.ChannelParams)), B, K,
                                          sh.size, sh.ctypes.data_as(C.POINTER(C.c_int32)), float(sampling_frequency),
                                          self._ptrs(outs_re), self._ptrs(outs_im), flags)
        self._check(rc, "gat_group_correlate")

    def gather(self, outs_re, outs_im, B: int, K: int, L: int, M: int) -> np.ndarray:
        """complex64 [B, K, L, M]: the members' outputs concatenated along the channel axis (synchronises)."""
        h_re = np.empty((B, K, L, M), dtype=np.float32)
        h_im = np.empty_like(h_re)
        rc = self.lib.gat_group_gather(self._h, self._ptrs(outs_re), self._ptrs(outs_im), B, K, L, M,
                                       h_re.ctypes.data_as(self._C.c_void_p), h_im.ctypes.data_as(self._C.c_void_p))
        self._check(rc, "gat_group_gather")
        return h_re + 1j * h_im

    def sync(self):
        self._check(self.lib.gat_group_sync(self._h), "gat_group_sync")
