"""The operator surface of the hot path, as the reference calls it through Tracking.jl:

* ``downconvert_and_correlate(system, signal, correlator, code_replica, code_phase,
  carrier_replica, carrier_phase, downconverted_signal, code_frequency,
  correlator_sample_shifts, carrier_frequency, sampling_frequency, signal_start_sample,
  num_samples, prn) -> correlator``                      (call site src/benchmarks.jl:63-79)
* ``gen_code_replica(code_replica, system, code_frequency, sampling_frequency, start_code_phase,
  start_sample, num_samples, correlator_sample_shifts, prn)``
                                                (scripts/code_replica_experiment.jl:70)
* ``StreamCorrelator`` -- the batched form (B consecutive integration blocks x K channels per
  launch) that a receiver and bench.py use; same arithmetic, one fused launch.

Same argument names and meaning as the reference; frequencies are plain floats in Hz (the
reference uses Unitful quantities), ``prn`` and ``signal_start_sample`` are 1-based as in Julia.
Everything executes in libgat (HIP); there is no CPU fallback.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .context import Context, get_context
from .correlator import EarlyPromptLateCorrelator
from .gen_signal import StructSignal, make_params
from .signals import GNSSSystem, get_code_frequency


_DTYPE_LAYOUT = {torch.float32: _lib.GAT_LAYOUT_INTERLEAVED, torch.int16: _lib.GAT_LAYOUT_INTERLEAVED_I16,
                 torch.int8: _lib.GAT_LAYOUT_INTERLEAVED_I8}


def _signal_desc(re: torch.Tensor, im: torch.Tensor | None, num_samples: int, start: int = 0,
                 block_stride: int | None = None, per_channel: bool = False) -> _lib.SignalDesc:
    """Describe a planar float32 pair [(K,) M, Ntot] or an interleaved tensor [(K,) M, Ntot, 2] of
    float32 (ComplexF32), int16 or int8 {re, im} pairs."""
    il = im is None
    t = re
    if not t.is_cuda:
        raise ValueError("signal must live on the HIP device")
    if il:
        if t.dtype not in _DTYPE_LAYOUT:
            raise ValueError("interleaved signal must be float32, int16 or int8")
        if t.shape[-1] != 2 or t.stride(-1) != 1 or t.stride(-2) != 2:
            raise ValueError("interleaved signal must be [..., N, 2] with unit inner strides")
        dims = t.dim() - 1
        stride = lambda d: t.stride(d - 1) // 2  # noqa: E731  (in complex samples)
        ntot = t.shape[-2]
        layout = _DTYPE_LAYOUT[t.dtype]
    else:
        if t.dtype != torch.float32 or im.dtype != torch.float32:
            raise ValueError("planar signal planes must be float32")
        if t.stride(-1) != 1 or im.stride() != t.stride() or im.shape != t.shape:
            raise ValueError("planar signal planes must be sample-contiguous with equal strides")
        dims = t.dim()
        stride = lambda d: t.stride(d)  # noqa: E731
        ntot = t.shape[-1]
        layout = _lib.GAT_LAYOUT_PLANAR
    if start < 0 or start + num_samples > ntot:
        raise ValueError("signal_start_sample/num_samples outside the signal")
    d = _lib.SignalDesc()
    off = start * (_lib.SAMPLE_BYTES[layout] if il else 4)
    d.re = t.data_ptr() + off
    d.im = None if il else im.data_ptr() + off
    d.layout = layout
    d.num_samples = num_samples
    d.num_ants = t.shape[dims - 2] if dims >= 2 else 1
    d.ant_stride = stride(-2) if dims >= 2 else ntot
    d.block_stride = num_samples if block_stride is None else block_stride
    d.chan_stride = stride(-3) if (per_channel and dims >= 3) else 0
    return d


def downconvert_and_correlate(system: GNSSSystem, signal: StructSignal, correlator: EarlyPromptLateCorrelator,
                              code_replica, code_phase: float, carrier_replica, carrier_phase: float,
                              downconverted_signal, code_frequency: float, correlator_sample_shifts,
                              carrier_frequency: float, sampling_frequency: float,
                              signal_start_sample: int, num_samples: int, prn: int,
                              flags: int = 0) -> EarlyPromptLateCorrelator:
    """Mirror of ``Tracking.downconvert_and_correlate!`` (src/benchmarks.jl:63-79).

    ``code_replica``, ``carrier_replica`` and ``downconverted_signal`` are the reference's scratch
    buffers; the fused kernel materialises none of them, so they are accepted and ignored.
    ``carrier_phase`` is in cycles (src/algorithms.jl:172).  Returns a NEW correlator whose
    accumulators are the result (the reference's functional update)."""
    ctx = get_context(signal.re.device)
    ctx.set_codes(system.codes)
    if signal.num_ants != correlator.num_ants:
        raise ValueError(f"signal has {signal.num_ants} antennas, correlator {correlator.num_ants}")
    shifts = np.ascontiguousarray(correlator_sample_shifts, dtype=np.int32)
    if shifts.size != correlator.num_accumulators:
        raise ValueError("one sample shift per accumulator is required")
    if not 1 <= prn <= system.codes.shape[0]:
        raise ValueError(f"prn {prn} outside 1..{system.codes.shape[0]}")
    desc = _signal_desc(signal.re, signal.im, num_samples, start=signal_start_sample - 1)
    prm = make_params(prn - 1, code_frequency, carrier_frequency, code_phase, carrier_phase, shape=(1, 1))
    L, M = shifts.size, correlator.num_ants
    out_re = torch.empty((L, M), dtype=torch.float32, device=ctx.device)
    out_im = torch.empty((L, M), dtype=torch.float32, device=ctx.device)
    ctx.downconvert_and_correlate(desc, prm, 1, 1, shifts, sampling_frequency, out_re, out_im, flags)
    return EarlyPromptLateCorrelator(M, L, _re=out_re, _im=out_im)


def gen_code_replica(code_replica: torch.Tensor, system: GNSSSystem, code_frequency: float,
                     sampling_frequency: float, start_code_phase: float, start_sample: int,
                     num_samples: int, correlator_sample_shifts, prn: int,
                     texture_coordinates: bool = False, texture_addressing=None) -> torch.Tensor:
    """Mirror of ``Tracking.gen_code_replica!`` (scripts/code_replica_experiment.jl:70) ==
    ``gen_code_replica_kernel!`` with ``latest_shift = shifts[1]`` in the 0-based convention of
    kernel 5431 (src/algorithms.jl:752-758): fills
    ``code_replica[start_sample-1 : start_sample-1 + num_samples + (shifts[-1]-shifts[0])]`` with
    ``c[floor(fc/fs*(i + shifts[0]) + phase) mod Lc]``.  ``texture_coordinates=True`` addresses the
    table through a Float32 normalised coordinate like ``gen_code_replica_texture_mem_kernel!``
    (src/algorithms.jl:121-140) -- for the code-phase-error study only; ``texture_addressing=(coord_frac_bits,
    texel_frac_bits)`` adds the fixed-point model of the texture unit's addressing (include/gat.h
    gat_gen_code_replica_texaddr)."""
    ctx = get_context(code_replica.device)
    ctx.set_codes(system.codes)
    shifts = np.asarray(correlator_sample_shifts, dtype=np.int64)
    count = int(num_samples + shifts[-1] - shifts[0])
    view = code_replica[start_sample - 1:]
    ctx.gen_code_replica(view, count, prn - 1, code_frequency, sampling_frequency, start_code_phase,
                         int(shifts[0]), f32_coordinates=texture_coordinates, texture_addressing=texture_addressing)
    return code_replica


def gen_code_replica_nsat(code_replica: torch.Tensor, system: GNSSSystem, code_frequency, sampling_frequency: float,
                          start_code_phase, num_samples: int, latest_shift: int, prns) -> torch.Tensor:
    """Mirror of ``gen_code_replica_texture_mem_strided_nsat_kernel!`` (src/algorithms.jl:78-98): the replicas of several
    satellites in one launch.  ``code_replica`` is float32 [num_sats, >= num_samples] (the reference's column-major
    [num_samples x num_sats]); row k gets ``c_k[floor(fc_k/fs*(i + latest_shift) + phase_k) mod Lc]`` for i = 0 ..
    num_samples-1 (0-based ``i``; the reference's ``thread_idx`` is 1-based).  ``prns`` are 1-based; ``code_frequency`` and
    ``start_code_phase`` may be scalars or one value per satellite."""
    ctx = get_context(code_replica.device)
    ctx.set_codes(system.codes)
    prns = np.atleast_1d(np.asarray(prns, dtype=np.int64))
    if (prns < 1).any() or (prns > system.codes.shape[0]).any():
        raise ValueError("prn outside the code table")
    k = prns.size
    prm = make_params(prns - 1, np.broadcast_to(np.asarray(code_frequency, dtype=np.float64), (k,)), 0.0,
                      np.broadcast_to(np.asarray(start_code_phase, dtype=np.float64), (k,)), 0.0, shape=(k,))
    ctx.gen_code_replica_multi(code_replica, int(num_samples), ctx.params_to_device(prm), k, sampling_frequency,
                               int(latest_shift))
    return code_replica


def downconvert_and_accumulate_strided(accum_re, accum_im, carrier_replica_re, carrier_replica_im,
                                       downconverted_signal_re, downconverted_signal_im, signal_re: torch.Tensor,
                                       signal_im: torch.Tensor, system: GNSSSystem, code_frequency: float,
                                       carrier_frequency: float, sampling_frequency: float, start_code_phase: float,
                                       carrier_phase: float, num_samples: int, correlator_sample_shifts, prn: int):
    """Mirror of ``downconvert_and_accumulate_strided_kernel!`` (src/algorithms.jl:828-866): the materialising middle
    stage of the reference's algorithm 2 (debug export; the fused correlator writes none of this).  Fills
    ``carrier_replica`` [N], ``downconverted_signal`` [M, N] and ``accum`` [L, M, N] (= the reference's column-major
    [N x M x L]); any of them may be None.  The code replica is evaluated in place (no replica buffer argument)."""
    ctx = get_context(signal_re.device)
    ctx.set_codes(system.codes)
    desc = _signal_desc(signal_re, signal_im, int(num_samples))
    prm = make_params(int(prn) - 1, code_frequency, carrier_frequency, start_code_phase, carrier_phase, shape=(1,))
    ctx.downconvert_and_accumulate(desc, prm, correlator_sample_shifts, sampling_frequency, carrier_replica_re,
                                   carrier_replica_im, downconverted_signal_re, downconverted_signal_im, accum_re, accum_im)


class StreamCorrelator:
    """Batched-stream operator: B consecutive integration blocks x K satellite channels per call.

    out[b, k, l, m] = sum_n x[n + b*N, m] * conj(carrier_{k,b}[n]) * code_{k,b}[n + shift_l]

    The per-(block, channel) parameters (what a tracking loop updates every block) live on the
    device; ``set_params`` uploads them.  ``__call__`` enqueues one fused launch on the context's
    stream and returns the (re, im) output tensors [B, K, L, M] without synchronising."""

    def __init__(self, system: GNSSSystem, num_samples: int, num_ants: int, num_blocks: int,
                 num_channels: int, correlator_sample_shifts, sampling_frequency: float,
                 flags: int = 0, per_channel_signal: bool = False, device=None, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else get_context(device)
        self.ctx.set_codes(system.codes)
        self.system = system
        self.N, self.M, self.B, self.K = int(num_samples), int(num_ants), int(num_blocks), int(num_channels)
        self.shifts = np.ascontiguousarray(correlator_sample_shifts, dtype=np.int32)
        self.L = int(self.shifts.size)
        self.fs = float(sampling_frequency)
        self.flags = int(flags)
        self.per_channel_signal = bool(per_channel_signal)
        dev = self.ctx.device
        self.out_re = torch.empty((self.B, self.K, self.L, self.M), dtype=torch.float32, device=dev)
        self.out_im = torch.empty_like(self.out_re)
        self.params_dev = None
        self._prepared = None

    def set_params(self, params: np.ndarray):
        params = np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE)
        if params.shape != (self.B, self.K):
            raise ValueError(f"params must be [{self.B}, {self.K}]")
        if (params["prn"] < 0).any() or (params["prn"] >= self.system.codes.shape[0]).any():
            raise ValueError("prn outside the code table")
        self.params_dev = self.ctx.params_to_device(params)

    def describe(self, re: torch.Tensor, im: torch.Tensor | None) -> _lib.SignalDesc:
        d = _signal_desc(re, im, self.N, per_channel=self.per_channel_signal)
        ntot = re.shape[-2] if im is None else re.shape[-1]
        if ntot < self.B * self.N:
            raise ValueError("signal shorter than num_blocks * num_samples")
        if d.num_ants != self.M:
            raise ValueError(f"signal has {d.num_ants} antennas, expected {self.M}")
        return d

    def launch(self, desc: _lib.SignalDesc):
        """Enqueue with a pre-built descriptor (lowest per-call overhead; used by bench.py)."""
        if self.params_dev is None:
            raise RuntimeError("set_params() has not been called")
        self.ctx.set_codes(self.system.codes)  # another operator may have bound its own table to this context
        key = (id(desc), self.params_dev.data_ptr())
        if self._prepared is None or self._prepared[0] != key:
            self._prepared = (key, self.ctx.prepared_call(desc, self.params_dev, self.B, self.K, self.shifts,
                                                          self.fs, self.out_re, self.out_im, self.flags))
        self._prepared[1]()

    def __call__(self, re: torch.Tensor, im: torch.Tensor | None = None):
        self.launch(self.describe(re, im))
        return self.out_re, self.out_im

    def result(self) -> np.ndarray:
        """complex64 [B, K, L, M] on the host (synchronises)."""
        if getattr(self.ctx, "own_stream", False):
            self.ctx.sync()  # the copies below are ordered with PyTorch's stream only
        return (self.out_re.cpu().numpy() + 1j * self.out_im.cpu().numpy()).astype(np.complex64)


def reduce_cplx_multi(in_re: torch.Tensor, in_im: torch.Tensor):
    """Column sums of a planar complex array -- the two-pass ``reduce_cplx_multi_3/4/5`` sequence
    of the reference (src/reduction.jl:93, :331, :548; src/algorithms.jl:914-922).  Input
    [..., n] float32 (C order: the reference's [n x M x L] column-major); returns (re, im) of
    shape [...]."""
    ctx = get_context(in_re.device)
    if in_re.shape != in_im.shape or in_re.dtype != torch.float32 or not in_re.is_contiguous() or not in_im.is_contiguous():
        raise ValueError("inputs must be contiguous float32 tensors of equal shape")
    n = in_re.shape[-1]
    cols = in_re.numel() // n
    out_re = torch.empty(in_re.shape[:-1], dtype=torch.float32, device=ctx.device)
    out_im = torch.empty_like(out_re)
    ctx.reduce_cplx_multi(in_re, in_im, n, cols, out_re, out_im)
    return out_re, out_im
