"""Correlator types of the operator surface (Tracking.jl names as the reference imports them:
``NumAnts``, ``NumAccumulators``, ``EarlyPromptLateCorrelator``, ``get_correlator_sample_shifts``;
src/GPUAcceleratedTracking.jl:22, src/benchmarks.jl:104-107)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .signals import GNSSSystem, get_code_frequency


@dataclass(frozen=True)
class NumAnts:
    value: int

    def __int__(self):
        return self.value


@dataclass(frozen=True)
class NumAccumulators:
    value: int

    def __int__(self):
        return self.value


def _as_int(x) -> int:
    return int(getattr(x, "value", x))


class EarlyPromptLateCorrelator:
    """``EarlyPromptLateCorrelator(NumAnts(M), NumAccumulators(L))``.

    Holds the L x M complex accumulators (antenna fastest: ``accum[antenna_idx, corr_idx]``,
    src/algorithms.jl:628).  After ``downconvert_and_correlate`` they live on the device;
    ``accumulators`` copies them to the host (synchronising)."""

    def __init__(self, num_ants=NumAnts(1), num_accumulators=NumAccumulators(3), _re=None, _im=None):
        self.num_ants = _as_int(num_ants)
        self.num_accumulators = _as_int(num_accumulators)
        if self.num_ants < 1 or self.num_accumulators < 1:
            raise ValueError("num_ants and num_accumulators must be positive")
        self._re = _re  # torch float32 [L, M] (device) or None
        self._im = _im

    @property
    def accumulators(self) -> np.ndarray:
        """complex64 [L, M]; zeros before the first correlation."""
        if self._re is None:
            return np.zeros((self.num_accumulators, self.num_ants), dtype=np.complex64)
        return (self._re.cpu().numpy() + 1j * self._im.cpu().numpy()).astype(np.complex64)

    @property
    def device_accumulators(self):
        """(re, im) float32 device tensors [L, M] (no synchronisation), or (None, None)."""
        return self._re, self._im


def get_accumulators(correlator: EarlyPromptLateCorrelator) -> np.ndarray:
    return correlator.accumulators


def get_num_ants(correlator: EarlyPromptLateCorrelator) -> int:
    return correlator.num_ants


def get_num_accumulators(correlator: EarlyPromptLateCorrelator) -> int:
    return correlator.num_accumulators


def get_correlator_sample_shifts(system: GNSSSystem, correlator: EarlyPromptLateCorrelator,
                                 sampling_frequency: float, preferred_code_shift: float = 0.5) -> np.ndarray:
    """``get_correlator_sample_shifts(system, correlator, fs, 0.5)`` (src/benchmarks.jl:105):
    s = max(1, round(shift*fs/fc)); taps (l - L//2)*s.  int32 [L]."""
    L = correlator.num_accumulators
    out = np.empty(L, dtype=np.int32)
    rc = _lib.load().gat_sample_shifts(L, float(sampling_frequency), float(get_code_frequency(system)),
                                       float(preferred_code_shift), out.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc != 0:
        raise _lib.GatError(rc, "gat_sample_shifts")
    return out
