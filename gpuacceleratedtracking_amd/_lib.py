"""ctypes binding of libgat.so -- the only way the Python host layer reaches the GPU.

There is NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

GAT_OK = 0
GAT_FLAG_ATOMIC = 1
GAT_FLAG_GRAPH = 2
GAT_LAYOUT_PLANAR = 0
GAT_LAYOUT_INTERLEAVED = 1
GAT_LAYOUT_INTERLEAVED_I16 = 2
GAT_LAYOUT_INTERLEAVED_I8 = 3
# bytes of one complex sample per layout
SAMPLE_BYTES = {0: 8, 1: 8, 2: 4, 3: 2}
GAT_MAX_TAPS = 32
# kernel selection (gat_set_matrix_core)
GAT_MC_VECTOR, GAT_MC_AUTO, GAT_MC_F32, GAT_MC_BF16_SPLIT = 0, 1, 2, 3

EXPORTS = [
    "gat_create", "gat_destroy", "gat_set_stream", "gat_sync", "gat_last_error", "gat_version",
    "gat_device_info", "gat_set_codes", "gat_gen_codes", "gat_sample_shifts",
    "gat_downconvert_and_correlate", "gat_downconvert_and_correlate_dev", "gat_gen_code_replica",
    "gat_gen_code_replica_f32coord",
    "gat_gen_signal", "gat_reduce_cplx_multi", "gat_tracking_update", "gat_malloc", "gat_free", "gat_memcpy_h2d",
    "gat_memcpy_d2h", "gat_memset", "gat_timer_start", "gat_timer_stop", "gat_last_launch_info", "gat_set_matrix_core", "gat_tracking_run", "gat_set_vector_tiling", "gat_gen_code_replica_multi",
    "gat_downconvert_and_accumulate", "gat_gen_signal_noisy", "gat_set_option",
    # several devices from one host thread (channel sharding, no collective)
    "gat_device_count", "gat_memcpy_peer", "gat_group_create", "gat_group_destroy", "gat_group_size", "gat_group_ctx",
    "gat_group_last_error", "gat_group_shard", "gat_group_set_codes", "gat_group_replicate", "gat_group_correlate",
    "gat_group_gather", "gat_group_sync",
    # resident correlator: single-block calls without a kernel launch
    "gat_resident_open", "gat_resident_correlate", "gat_resident_info_get", "gat_resident_park", "gat_resident_close",
    "gat_tracking_update_host", "gat_resident_tracking_run", "gat_resident_park_all",
    # measurement: per-call statistics, the in-run read ceiling; the texture-addressing study
    "gat_timer_lap", "gat_timer_laps", "gat_debug_read_stream", "gat_gen_code_replica_texaddr",
]


class GatError(RuntimeError):
    """A libgat call returned a non-zero status."""

    def __init__(self, status: int, where: str, message: str = ""):
        self.status = status
        kind = {1: "GAT_ERR_ARG", 2: "GAT_ERR_RANGE", 3: "GAT_ERR_STATE", 4: "GAT_ERR_UNSUPPORTED",
                5: "GAT_ERR_NOMEM"}.get(status, f"hipError {-status}" if status < 0 else str(status))
        super().__init__(f"{where}: {kind}" + (f" ({message})" if message else ""))


class ChannelParams(C.Structure):
    """gat_channel_params (include/gat.h)."""

    _fields_ = [("prn", C.c_int32), ("reserved", C.c_int32), ("code_freq_hz", C.c_double),
                ("carrier_freq_hz", C.c_double), ("code_phase_chips", C.c_double),
                ("carrier_phase_cycles", C.c_double)]


PARAMS_DTYPE = np.dtype([("prn", "<i4"), ("reserved", "<i4"), ("code_freq_hz", "<f8"),
                         ("carrier_freq_hz", "<f8"), ("code_phase_chips", "<f8"),
                         ("carrier_phase_cycles", "<f8")])
assert PARAMS_DTYPE.itemsize == C.sizeof(ChannelParams) == 40


class SignalDesc(C.Structure):
    """gat_signal_desc (include/gat.h)."""

    _fields_ = [("re", C.c_void_p), ("im", C.c_void_p), ("layout", C.c_int32),
                ("num_ants", C.c_int32), ("num_samples", C.c_int64), ("ant_stride", C.c_int64),
                ("block_stride", C.c_int64), ("chan_stride", C.c_int64)]


class LoopConfig(C.Structure):
    """gat_loop_config (include/gat.h)."""

    _fields_ = [("block_seconds", C.c_double), ("pll_bandwidth_hz", C.c_double), ("dll_bandwidth_hz", C.c_double),
                ("code_freq_nominal_hz", C.c_double), ("carrier_center_hz", C.c_double), ("if_hz", C.c_double),
                ("early_late_spacing_chips", C.c_double), ("code_length", C.c_int32), ("num_taps", C.c_int32),
                ("early_index", C.c_int32), ("prompt_index", C.c_int32), ("late_index", C.c_int32)]


LOOP_STATE_DTYPE = np.dtype([(n, "<f8") for n in (
    "init_carrier_doppler_hz", "carrier_doppler_hz", "code_doppler_hz", "pll_acc1", "pll_acc2", "dll_acc",
    "last_pll_error_cycles", "last_dll_error_chips", "prompt_power")])


class LaunchInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("workgroups", "threads", "splits", "ant_tile", "vec",
                                          "lds_bytes", "finalize_launched", "matrix_core", "channels_per_wg",
                                          "blocks_per_wg", "prefetch_depth", "bf16_terms")]


class ResidentConfig(C.Structure):
    """gat_resident_config (include/gat.h)."""

    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "idle_us", "life_ms", "max_calls", "max_workgroups", "host_pollers", "doorbell")]


class ResidentInfo(C.Structure):
    """gat_resident_info (include/gat.h)."""

    _fields_ = [("workgroups", C.c_int32), ("splits", C.c_int32), ("running", C.c_int32), ("last_exit", C.c_int32),
                ("launches", C.c_uint64), ("calls", C.c_uint64)]


_LIB = None


def library_path() -> str:
    """Path of libgat.so; GAT_LIBRARY overrides it (kernel experiments: A/B builds)."""
    return os.environ.get("GAT_LIBRARY") or _build.LIB


def load(build_if_missing: bool = True):
    """Load libgat.so (building it with hipcc when absent/stale and hipcc is available)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if build_if_missing and not os.environ.get("GAT_LIBRARY") and _build.is_stale():
        try:
            _build.build_libgat()
        except Exception as exc:  # no hipcc on this machine
            if not os.path.exists(path):
                raise ImportError(f"libgat.so is not built and hipcc failed: {exc}") from exc
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `python -m gpuacceleratedtracking_amd.build`")
    lib = C.CDLL(path)
    vp, i32, i64, u32, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double
    pp = C.POINTER(ChannelParams)
    sp = C.POINTER(SignalDesc)
    i32p = C.POINTER(C.c_int32)
    sigs = {
        "gat_create": (i32, [i32, vp, C.POINTER(vp)]),
        "gat_destroy": (i32, [vp]),
        "gat_set_stream": (i32, [vp, vp]),
        "gat_sync": (i32, [vp]),
        "gat_last_error": (C.c_char_p, [vp]),
        "gat_version": (C.c_char_p, []),
        "gat_device_info": (i32, [vp, C.c_char_p, C.c_size_t, i32p, i32p]),
        "gat_set_codes": (i32, [vp, C.POINTER(C.c_int8), i32, i32]),
        "gat_gen_codes": (i32, [C.c_char_p, i32, C.POINTER(C.c_int8), i32p, C.POINTER(dbl)]),
        "gat_sample_shifts": (i32, [i32, dbl, dbl, dbl, i32p]),
        "gat_downconvert_and_correlate": (i32, [vp, sp, pp, i32, i32, i32, i32p, dbl, vp, vp, u32]),
        "gat_downconvert_and_correlate_dev": (i32, [vp, sp, vp, i32, i32, i32, i32p, dbl, vp, vp, u32]),
        "gat_gen_code_replica": (i32, [vp, vp, i64, i32, dbl, dbl, dbl, i64]),
        "gat_gen_code_replica_f32coord": (i32, [vp, vp, i64, i32, dbl, dbl, dbl, i64]),
        "gat_gen_signal": (i32, [vp, vp, vp, i32, i64, i32, i64, i64, i32, i32, vp, dbl, dbl]),
        "gat_reduce_cplx_multi": (i32, [vp, vp, vp, i64, i32, vp, vp]),
        "gat_tracking_update": (i32, [vp, vp, vp, i32, i32, C.POINTER(LoopConfig), vp, vp, vp]),
        "gat_tracking_run": (i32, [vp, C.POINTER(SignalDesc), i32, i32, i32, C.POINTER(C.c_int32), dbl,
                                   C.POINTER(LoopConfig), vp, vp, vp, vp, vp, i64, C.c_uint32, C.POINTER(C.c_int32)]),
        "gat_malloc": (i32, [vp, C.c_size_t, C.POINTER(vp)]),
        "gat_free": (i32, [vp, vp]),
        "gat_memcpy_h2d": (i32, [vp, vp, vp, C.c_size_t]),
        "gat_memcpy_d2h": (i32, [vp, vp, vp, C.c_size_t]),
        "gat_memset": (i32, [vp, vp, i32, C.c_size_t]),
        "gat_timer_start": (i32, [vp]),
        "gat_timer_stop": (i32, [vp, C.POINTER(C.c_float)]),
        "gat_last_launch_info": (i32, [vp, C.POINTER(LaunchInfo), C.c_size_t]),
        "gat_set_matrix_core": (i32, [vp, i32]),
        "gat_set_vector_tiling": (i32, [vp, i32, i32, i32]),
        "gat_set_option": (i32, [vp, C.c_char_p, i64]),
        "gat_gen_code_replica_multi": (i32, [vp, vp, i64, i64, i32, vp, dbl, i64]),
        "gat_downconvert_and_accumulate": (i32, [vp, sp, pp, i32, i32p, dbl, vp, vp, vp, vp, vp, vp]),
        "gat_gen_signal_noisy": (i32, [vp, vp, vp, i32, i64, i32, i64, i64, i32, i32, vp, dbl, dbl, vp, dbl, C.c_uint64]),
        "gat_device_count": (i32, [i32p]),
        "gat_memcpy_peer": (i32, [vp, vp, vp, vp, C.c_size_t]),
        "gat_group_create": (i32, [i32, i32p, C.POINTER(vp)]),
        "gat_group_destroy": (i32, [vp]),
        "gat_group_size": (i32, [vp, i32p]),
        "gat_group_ctx": (i32, [vp, i32, C.POINTER(vp)]),
        "gat_group_last_error": (C.c_char_p, [vp]),
        "gat_group_shard": (i32, [vp, i32, i32, i32p, i32p]),
        "gat_group_set_codes": (i32, [vp, C.POINTER(C.c_int8), i32, i32]),
        "gat_group_replicate": (i32, [vp, i32, C.POINTER(vp), C.c_size_t]),
        "gat_group_correlate": (i32, [vp, sp, pp, i32, i32, i32, i32p, dbl, C.POINTER(vp), C.POINTER(vp), u32]),
        "gat_group_gather": (i32, [vp, C.POINTER(vp), C.POINTER(vp), i32, i32, i32, i32, vp, vp]),
        "gat_group_sync": (i32, [vp]),
        "gat_resident_open": (i32, [vp, sp, i32, i32, i32p, dbl, C.POINTER(ResidentConfig), C.POINTER(vp)]),
        "gat_resident_correlate": (i32, [vp, pp, i64, vp, vp]),
        "gat_resident_info_get": (i32, [vp, C.POINTER(ResidentInfo), C.c_size_t]),
        "gat_resident_park": (i32, [vp]),
        "gat_resident_close": (i32, [vp]),
        "gat_tracking_update_host": (i32, [vp, vp, i32, i32, C.POINTER(LoopConfig), vp, vp, vp]),
        "gat_resident_tracking_run": (i32, [vp, i32, i64, i64, C.POINTER(LoopConfig), vp, vp, vp, vp, i64]),
        "gat_resident_park_all": (i32, [vp]),
        "gat_timer_lap": (i32, [vp]),
        "gat_timer_laps": (i32, [vp, C.POINTER(C.c_float), i32, i32p]),
        "gat_debug_read_stream": (i32, [vp, vp, C.c_size_t, i32, i32, C.POINTER(C.c_float)]),
        "gat_gen_code_replica_texaddr": (i32, [vp, vp, i64, i32, dbl, dbl, dbl, i64, i32, i32]),
    }
    assert sorted(sigs) == sorted(EXPORTS)
    for name, (res, args) in sigs.items():
        if os.environ.get("GAT_LIBRARY") and not hasattr(lib, name):
            continue  # A/B against an OLDER development build that predates a symbol (GAT_LIBRARY override only)
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
