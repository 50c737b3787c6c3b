"""GNSS system objects: stand-ins for ``GNSSSignals.GPSL1`` / ``GPSL5`` as the reference uses
them (``system.codes``, ``get_code_frequency``, ``get_code_length``; src/benchmarks.jl:43-48,
src/gen_signal.jl:64-65, src/GPUAcceleratedTracking.jl:39-42).  Code tables come from libgat's
host-side generators (gat_gen_codes)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def generate_codes(system_name: str, num_prns: int = 32) -> tuple[np.ndarray, float]:
    """int8 +-1 table [num_prns, code_length] and the nominal code frequency in Hz."""
    lib = _lib.load()
    lc, fc = C.c_int32(), C.c_double()
    rc = lib.gat_gen_codes(system_name.encode(), 0, None, C.byref(lc), C.byref(fc))
    if rc != 0:
        raise _lib.GatError(rc, "gat_gen_codes")
    out = np.empty((num_prns, lc.value), dtype=np.int8)
    rc = lib.gat_gen_codes(system_name.encode(), num_prns, out.ctypes.data_as(C.POINTER(C.c_int8)),
                           C.byref(lc), C.byref(fc))
    if rc != 0:
        raise _lib.GatError(rc, "gat_gen_codes")
    return out, fc.value


class GNSSSystem:
    """``system`` argument of the operator surface."""

    name = ""

    def __init__(self, use_gpu: bool = True, num_prns: int = 32, codes: np.ndarray | None = None,
                 code_frequency: float | None = None):
        if codes is None:
            codes, fc = generate_codes(self.name, num_prns)
        else:
            codes = np.ascontiguousarray(codes, dtype=np.int8)
            fc = code_frequency
        self.codes = codes                      # [P, Lc] int8 (reference: codes[chip, prn])
        self.code_frequency = float(code_frequency if code_frequency is not None else fc)
        self.use_gpu = bool(getattr(use_gpu, "value", use_gpu))

    @property
    def code_length(self) -> int:
        return int(self.codes.shape[1])


class GPSL1(GNSSSystem):
    name = "GPSL1"


class GPSL5(GNSSSystem):
    name = "GPSL5"


def get_code_frequency(system: GNSSSystem) -> float:
    return system.code_frequency


def get_code_length(system: GNSSSystem) -> int:
    return system.code_length


GNSSDICT = {"GPSL1": GPSL1, "GPSL5": GPSL5}  # src/GPUAcceleratedTracking.jl:39-42
