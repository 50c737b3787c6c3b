"""Synthetic IF signal generation on the device: mirror of src/gen_signal.jl.

``gen_signal(system, prn, carrier_frequency, num_samples; num_ants, duration, start_code_phase,
start_carrier_phase) -> (signal, sampling_frequency)`` (src/gen_signal.jl:1-25).  The signal is a
planar complex array like the reference's ``StructArray{ComplexF32}``: ``signal.re`` /
``signal.im`` are float32 tensors of shape [M, N] in C order, i.e. column-major [N x M]
(sample fastest, src/gen_signal.jl:179)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .context import get_context
from .correlator import _as_int
from .signals import GNSSSystem, get_code_frequency


class StructSignal:
    """Planar complex signal on the device.  re/im: float32 [..., M, N] (C order)."""

    def __init__(self, re: torch.Tensor, im: torch.Tensor):
        if re.shape != im.shape or re.dtype != torch.float32 or im.dtype != torch.float32:
            raise ValueError("re/im must be float32 tensors of equal shape")
        self.re = re
        self.im = im

    @property
    def num_samples(self) -> int:
        return int(self.re.shape[-1])

    @property
    def num_ants(self) -> int:
        return int(self.re.shape[-2]) if self.re.dim() >= 2 else 1

    def cpu(self):
        return self.re.cpu().numpy(), self.im.cpu().numpy()


def gen_blank_signal(system: GNSSSystem, num_samples: int, num_ants=1, prns=None, device=None) -> StructSignal:
    """``gen_blank_signal`` (src/gen_signal.jl:177-184): zero-filled planar signal."""
    ctx = get_context(device)
    m = _as_int(num_ants)
    shape = (m, num_samples) if prns is None else (len(prns), m, num_samples)
    return StructSignal(torch.zeros(shape, dtype=torch.float32, device=ctx.device),
                        torch.zeros(shape, dtype=torch.float32, device=ctx.device))


def make_params(prn, code_frequency, carrier_frequency, code_phase, carrier_phase, shape=None) -> np.ndarray:
    """Structured numpy array of ``gat_channel_params``; scalars broadcast.  ``prn`` is 0-based."""
    arrs = np.broadcast_arrays(np.asarray(prn), np.asarray(code_frequency, dtype=np.float64),
                               np.asarray(carrier_frequency, dtype=np.float64),
                               np.asarray(code_phase, dtype=np.float64),
                               np.asarray(carrier_phase, dtype=np.float64))
    if shape is not None:
        arrs = [np.broadcast_to(a, shape) for a in arrs]
    out = np.zeros(arrs[0].shape, dtype=_lib.PARAMS_DTYPE)
    out["prn"], out["code_freq_hz"], out["carrier_freq_hz"] = arrs[0], arrs[1], arrs[2]
    out["code_phase_chips"], out["carrier_phase_cycles"] = arrs[3], arrs[4]
    return out


_LAYOUT_DTYPE = {_lib.GAT_LAYOUT_INTERLEAVED: torch.float32, _lib.GAT_LAYOUT_INTERLEAVED_I16: torch.int16,
                 _lib.GAT_LAYOUT_INTERLEAVED_I8: torch.int8}


def gen_signal_stream(system: GNSSSystem, params: np.ndarray, sampling_frequency: float, num_samples: int,
                      num_ants: int = 1, layout: int = _lib.GAT_LAYOUT_PLANAR, device=None,
                      amplitude: float = 1.0, ant_pad: int = 0, steering_cycles=None, noise_sigma: float = 0.0,
                      seed: int = 0):
    """Batched form used by the stream benchmark: ``params`` is [B, K] (carrier phase in
    RADIANS, as ``start_carrier_phase`` in src/gen_signal.jl:88); block b holds the sum of its K
    channels times ``amplitude``.  Returns tensors: planar (re [M, B*N], im [M, B*N]) float32;
    interleaved layouts (x [M, B*N, 2], None) of float32 / int16 / int8 (integer layouts store
    ``rint(amplitude * x)`` saturated -- what an ADC front-end delivers).  ``ant_pad``: extra samples between
    the antennas' streams (the returned tensors are views [M, B*N(, 2)] of a padded allocation): antenna planes
    whose distance is a large power-of-two multiple collide on the same memory channels.  ``steering_cycles`` (M phases in
    cycles), ``noise_sigma``, ``seed``: per-antenna steering and complex white Gaussian noise (gat_gen_signal_noisy; the
    reference's generator is noise-free with identical antennas)."""
    ctx = get_context(device)
    ctx.set_codes(system.codes)
    params = np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE)
    if params.ndim != 2:
        raise ValueError("params must be [B, K]")
    B, K = params.shape
    m = _as_int(num_ants)
    dparams = ctx.params_to_device(params)
    row = B * num_samples + int(ant_pad)
    if layout == _lib.GAT_LAYOUT_PLANAR:
        re = torch.empty((m, row), dtype=torch.float32, device=ctx.device)[:, :B * num_samples]
        im = torch.empty((m, row), dtype=torch.float32, device=ctx.device)[:, :B * num_samples]
    else:
        re = torch.empty((m, row, 2), dtype=_LAYOUT_DTYPE[layout], device=ctx.device)[:, :B * num_samples]
        im = None
    steer = None
    if steering_cycles is not None:
        steer = torch.as_tensor(np.ascontiguousarray(steering_cycles, dtype=np.float32)).to(ctx.device)
    ctx.gen_signal(re, im, layout, num_samples, m, row, num_samples, B, K, dparams,
                   sampling_frequency, amplitude, steering_cycles=steer, noise_sigma=noise_sigma, seed=seed)
    return re, im


def gen_signal(system: GNSSSystem, prn, carrier_frequency: float, num_samples: int, num_ants=1,
               duration: float = 1e-3, start_code_phase: float = 0.0, start_carrier_phase: float = 0.0,
               device=None):
    """Mirror of ``gen_signal`` (src/gen_signal.jl:1-51).  ``prn`` is 1-based as in the reference;
    a list of PRNs yields one signal per satellite, shape [K, M, N] (src/gen_signal.jl:27-51,
    :155-175).  Returns ``(StructSignal, sampling_frequency)``."""
    fs = num_samples / duration  # src/gen_signal.jl:11
    m = _as_int(num_ants)
    prns = list(prn) if isinstance(prn, (list, tuple, np.ndarray)) else None
    fc = get_code_frequency(system)
    if prns is None:
        prm = make_params(int(prn) - 1, fc, carrier_frequency, start_code_phase, start_carrier_phase, shape=(1, 1))
        re, im = gen_signal_stream(system, prm, fs, num_samples, m, device=device)
        return StructSignal(re, im), fs
    res, ims = [], []
    for p in prns:
        prm = make_params(int(p) - 1, fc, carrier_frequency, start_code_phase, start_carrier_phase, shape=(1, 1))
        re, im = gen_signal_stream(system, prm, fs, num_samples, m, device=device)
        res.append(re)
        ims.append(im)
    return StructSignal(torch.stack(res), torch.stack(ims)), fs
