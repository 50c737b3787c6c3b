"""Device context: one libgat ctx per (device, HIP stream).

PyTorch is plumbing here: it owns device memory (tensors) and streams; every compute call goes
through the C ABI with raw device pointers.  Mirrors the implicit CUDA.jl device/stream state of
the reference (``CUDA.@sync`` at src/benchmarks.jl:120).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GatError


def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


class Context:
    """Owns a ``gat_ctx``.  Not thread-safe (as the C ABI states); distinct contexts are."""

    def __init__(self, device: int | torch.device | None = None, stream: torch.cuda.Stream | str | None = None):
        """``stream``: a torch stream (default: the current one) -- or ``"own"``: the library creates and owns a
        non-blocking stream (GAT_OWN_STREAM).  Small launches on an owned stream end with a completion flag in pinned host
        memory and ``sync()`` spins on it (a single-block call + sync is ~6 us shorter than through hipStreamSynchronize);
        work on such a context is NOT ordered with PyTorch's streams: the caller synchronises around it
        (``torch.cuda.synchronize()`` after producing the inputs, ``ctx.sync()`` before reading the outputs)."""
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("gpuacceleratedtracking_amd needs a HIP device (no CPU fallback exists)")
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else (device.index or 0))
        self.own_stream = isinstance(stream, str)
        if self.own_stream and stream != "own":
            raise ValueError("stream must be a torch.cuda.Stream, None or 'own'")
        with torch.cuda.device(self.device):
            self.stream = None if self.own_stream else (stream if stream is not None else torch.cuda.current_stream(self.device))
        self._h = C.c_void_p()
        handle = C.c_void_p(-1) if self.own_stream else C.c_void_p(self.stream.cuda_stream)  # GAT_OWN_STREAM = (void *)-1
        rc = self.lib.gat_create(self.device.index, handle, C.byref(self._h))
        if rc != 0:
            raise GatError(rc, "gat_create")
        self._codes_key = None
        self._codes_obj = None

    # -- plumbing -----------------------------------------------------------------------
    def check(self, rc: int, where: str):
        if rc != 0:
            msg = self.lib.gat_last_error(self._h)
            raise GatError(rc, where, msg.decode() if msg else "")

    def close(self):
        if self._h:
            self.lib.gat_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self.check(self.lib.gat_sync(self._h), "gat_sync")

    def device_info(self) -> dict:
        buf = C.create_string_buffer(256)
        ver, cus = C.c_int32(), C.c_int32()
        self.check(self.lib.gat_device_info(self._h, buf, 256, C.byref(ver), C.byref(cus)), "gat_device_info")
        return {"name": buf.value.decode(), "hip_runtime": ver.value, "num_cus": cus.value}

    def last_launch_info(self) -> dict:
        info = _lib.LaunchInfo()
        self.check(self.lib.gat_last_launch_info(self._h, C.byref(info), C.sizeof(info)), "gat_last_launch_info")
        return {n: getattr(info, n) for n, _ in info._fields_}

    def set_matrix_core(self, mode):
        """Kernel selection for antenna-rich shapes (include/gat.h GAT_MC_*): True / 1 = auto (default:
        split-bf16 MFMA kernel where it applies), False / 0 = vector kernel only, 2 = f32-MFMA kernel,
        3 = split-bf16 MFMA kernel only."""
        self.check(self.lib.gat_set_matrix_core(self._h, int(mode)), "gat_set_matrix_core")

    def set_vector_tiling(self, max_antenna_tiles: int = 0, max_channels: int = 0, max_blocks: int = 0):
        """Caps on the vector kernel's workgroup tiling (include/gat.h gat_set_vector_tiling; 0 = unchanged)."""
        self.check(self.lib.gat_set_vector_tiling(self._h, int(max_antenna_tiles), int(max_channels), int(max_blocks)),
                   "gat_set_vector_tiling")

    def set_option(self, name: str, value: int):
        """Launch-geometry option by name (include/gat.h gat_set_option): tests force a code path on a small case, A/B
        runs compare geometries; never changes a result beyond summation order."""
        self.check(self.lib.gat_set_option(self._h, name.encode(), int(value)), f"gat_set_option({name})")

    def timer_start(self):
        self.check(self.lib.gat_timer_start(self._h), "gat_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        self.check(self.lib.gat_timer_stop(self._h, C.byref(ms)), "gat_timer_stop")
        return float(ms.value)

    def timer_lap(self):
        """One event on the context's stream (include/gat.h gat_timer_lap): before the first timed launch and after every one."""
        self.check(self.lib.gat_timer_lap(self._h), "gat_timer_lap")

    def timer_laps(self, capacity: int = 1 << 16) -> np.ndarray:
        """Waits for the newest lap; the intervals between consecutive laps in milliseconds (BenchmarkTools keeps every
        sample's time the same way, src/benchmarks.jl:1-9)."""
        buf = np.zeros(int(capacity), dtype=np.float32)
        n = C.c_int32()
        self.check(self.lib.gat_timer_laps(self._h, buf.ctypes.data_as(C.POINTER(C.c_float)), int(capacity), C.byref(n)), "gat_timer_laps")
        return buf[:n.value].astype(np.float64)

    def read_stream_ms(self, tensor_or_ptr, nbytes: int, variant: int = 0, launches: int = 8) -> np.ndarray:
        """A kernel that only reads ``nbytes`` of device memory, ``launches`` times; milliseconds per launch
        (include/gat.h gat_debug_read_stream: the in-run read ceiling of SURVEY section 8-d)."""
        ptr = tensor_or_ptr if isinstance(tensor_or_ptr, int) else _ptr(tensor_or_ptr)
        ms = np.zeros(int(launches), dtype=np.float32)
        self.check(self.lib.gat_debug_read_stream(self._h, C.c_void_p(ptr), int(nbytes), int(variant), int(launches),
                                                  ms.ctypes.data_as(C.POINTER(C.c_float))), "gat_debug_read_stream")
        return ms.astype(np.float64)

    def park_residents(self):
        """Ask every resident correlator of this context to leave the device and wait until it has (gat_resident_park_all):
        before anything that waits for the WHOLE device -- ``torch.cuda.synchronize()``, ``hipDeviceSynchronize`` -- which
        would otherwise sit out the resident kernels' idle limit (5 ms by default).  The next call starts them again."""
        self.check(self.lib.gat_resident_park_all(self._h), "gat_resident_park_all")

    def device_synchronize(self):
        """``torch.cuda.synchronize()`` that does not stall behind this context's resident correlators."""
        self.park_residents()
        torch.cuda.synchronize(self.device)

    # -- code tables --------------------------------------------------------------------
    def set_codes(self, codes: np.ndarray):
        """codes: int8 [num_prns, code_length] (C-order == reference's column-major [Lc x P]).

        The table is per-context state in libgat while the reference passes `system` with every call, and one context
        serves every operator on a (device, stream): each operator therefore re-binds its own table before every launch
        (a dual-frequency receiver alternates L1 and L5 on one context).  Cheap: the table bound last is remembered by
        IDENTITY -- a code table is immutable once it has been bound (as ``system.codes`` is in the reference: a constant of
        the GNSS system); to change chips, bind a new array object.  Only a different object is hashed, and only
        different contents are uploaded."""
        if codes is self._codes_obj:
            return
        arr = np.ascontiguousarray(codes, dtype=np.int8)
        key = (arr.shape, hash(arr.tobytes()))
        if key != self._codes_key:
            p, lc = arr.shape
            self.check(self.lib.gat_set_codes(self._h, arr.ctypes.data_as(C.POINTER(C.c_int8)), lc, p), "gat_set_codes")
            self._codes_key = key
        self._codes_obj = codes  # keeps the array alive, so the identity test above cannot be fooled by a recycled id
        # the identity shortcut above is only sound while the bound array does not change: make an in-place edit RAISE instead
        # of silently keeping the old chips on the device (bind a new array, or call invalidate_codes(), to change chips)
        if isinstance(codes, np.ndarray):
            try:
                codes.flags.writeable = False
            except ValueError:  # pragma: no cover - an array that does not own its flags
                pass

    def invalidate_codes(self):
        """Forget which table is bound: the next operator call hashes and, if the contents changed, uploads its table again
        (for a caller that had to edit a bound code array in place, after setting ``codes.flags.writeable = True``)."""
        self._codes_obj = None
        self._codes_key = None

    # -- operators ----------------------------------------------------------------------
    def downconvert_and_correlate(self, desc: _lib.SignalDesc, params, num_blocks: int, num_channels: int,
                                  shifts, sampling_frequency: float, out_re: torch.Tensor,
                                  out_im: torch.Tensor, flags: int = 0):
        """params: numpy structured array (host path) or a torch uint8/int tensor holding
        gat_channel_params on the device (device path)."""
        sh = np.ascontiguousarray(shifts, dtype=np.int32)
        shp = sh.ctypes.data_as(C.POINTER(C.c_int32))
        need = num_blocks * num_channels * sh.size * desc.num_ants
        if out_re.numel() < need or out_im.numel() < need or out_re.dtype != torch.float32:
            raise ValueError("output tensors too small or not float32")
        if isinstance(params, torch.Tensor):
            if params.numel() * params.element_size() < num_blocks * num_channels * 40:
                raise ValueError("device params tensor too small")
            rc = self.lib.gat_downconvert_and_correlate_dev(
                self._h, C.byref(desc), C.c_void_p(_ptr(params)), num_blocks, num_channels, sh.size, shp,
                float(sampling_frequency), C.c_void_p(_ptr(out_re)), C.c_void_p(_ptr(out_im)), flags)
            self.check(rc, "gat_downconvert_and_correlate_dev")
        else:
            prm = np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE)
            if prm.size != num_blocks * num_channels:
                raise ValueError("params must hold num_blocks * num_channels entries")
            rc = self.lib.gat_downconvert_and_correlate(
                self._h, C.byref(desc), prm.ctypes.data_as(C.POINTER(_lib.ChannelParams)), num_blocks,
                num_channels, sh.size, shp, float(sampling_frequency), C.c_void_p(_ptr(out_re)),
                C.c_void_p(_ptr(out_im)), flags)
            self.check(rc, "gat_downconvert_and_correlate")

    def prepared_call(self, desc: _lib.SignalDesc, params_dev: torch.Tensor, num_blocks: int, num_channels: int,
                      shifts, sampling_frequency: float, out_re: torch.Tensor, out_im: torch.Tensor, flags: int = 0):
        """Validate once and return a zero-argument callable that enqueues
        gat_downconvert_and_correlate_dev with pre-converted ctypes arguments (a tracking loop calls
        the same operator thousands of times per second; the reference pays Julia dispatch +
        `@cuda` argument conversion per launch, src/algorithms.jl:896)."""
        sh = np.ascontiguousarray(shifts, dtype=np.int32)
        need = num_blocks * num_channels * sh.size * desc.num_ants
        if out_re.numel() < need or out_im.numel() < need or out_re.dtype != torch.float32:
            raise ValueError("output tensors too small or not float32")
        if params_dev.numel() * params_dev.element_size() < num_blocks * num_channels * 40:
            raise ValueError("device params tensor too small")
        fn = self.lib.gat_downconvert_and_correlate_dev
        args = (self._h, C.byref(desc), C.c_void_p(_ptr(params_dev)), num_blocks, num_channels, int(sh.size),
                sh.ctypes.data_as(C.POINTER(C.c_int32)), float(sampling_frequency), C.c_void_p(_ptr(out_re)),
                C.c_void_p(_ptr(out_im)), int(flags))
        keep = (desc, sh, params_dev, out_re, out_im)  # keep the buffers alive with the closure

        def call(_fn=fn, _args=args, _keep=keep, _check=self.check):
            rc = _fn(*_args)
            if rc != 0:
                _check(rc, "gat_downconvert_and_correlate_dev")

        return call

    def gen_code_replica(self, out: torch.Tensor, count: int, prn: int, code_frequency: float,
                         sampling_frequency: float, code_phase: float, first_shift: int,
                         f32_coordinates: bool = False, texture_addressing: tuple[int, int] | None = None):
        """``texture_addressing = (coord_frac_bits, texel_frac_bits)``: the fixed-point model of the texture unit's
        addressing (include/gat.h gat_gen_code_replica_texaddr; study use)."""
        if out.numel() < count or out.dtype != torch.float32:
            raise ValueError("replica tensor too small or not float32")
        if texture_addressing is not None:
            cb, tb = texture_addressing
            rc = self.lib.gat_gen_code_replica_texaddr(self._h, C.c_void_p(_ptr(out)), count, prn, float(code_frequency),
                                                       float(sampling_frequency), float(code_phase), int(first_shift), int(cb), int(tb))
            self.check(rc, "gat_gen_code_replica_texaddr")
            return
        fn = self.lib.gat_gen_code_replica_f32coord if f32_coordinates else self.lib.gat_gen_code_replica
        rc = fn(self._h, C.c_void_p(_ptr(out)), count, prn, float(code_frequency), float(sampling_frequency),
                float(code_phase), int(first_shift))
        self.check(rc, "gat_gen_code_replica")

    def gen_code_replica_multi(self, out: torch.Tensor, count: int, params_dev: torch.Tensor, num_channels: int,
                               sampling_frequency: float, first_shift: int):
        """out: float32 [num_channels, >= count] (row k = channel k); params_dev: gat_channel_params[num_channels]."""
        if out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] < num_channels or out.shape[1] < count or out.stride(1) != 1:
            raise ValueError("replica tensor must be float32 [num_channels, >= count] with unit inner stride")
        if params_dev.numel() * params_dev.element_size() < num_channels * 40:
            raise ValueError("device params tensor too small")
        rc = self.lib.gat_gen_code_replica_multi(self._h, C.c_void_p(_ptr(out)), count, out.stride(0), num_channels,
                                                 C.c_void_p(_ptr(params_dev)), float(sampling_frequency), int(first_shift))
        self.check(rc, "gat_gen_code_replica_multi")

    def downconvert_and_accumulate(self, desc: _lib.SignalDesc, params, shifts, sampling_frequency: float,
                                   carrier_re=None, carrier_im=None, dw_re=None, dw_im=None, accum_re=None, accum_im=None):
        """The reference's materialising algorithm-2 stage (include/gat.h gat_downconvert_and_accumulate): any of
        carrier [N], downconverted [M, N], accum [L, M, N] (float32, contiguous) may be None."""
        sh = np.ascontiguousarray(shifts, dtype=np.int32)
        prm = np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE).reshape(-1)
        if prm.size != 1:
            raise ValueError("one channel")
        n, m = int(desc.num_samples), int(desc.num_ants)
        for t, need in ((carrier_re, n), (carrier_im, n), (dw_re, n * m), (dw_im, n * m), (accum_re, n * m * sh.size),
                        (accum_im, n * m * sh.size)):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < need):
                raise ValueError("debug outputs must be contiguous float32 tensors of sufficient size")
        rc = self.lib.gat_downconvert_and_accumulate(
            self._h, C.byref(desc), prm.ctypes.data_as(C.POINTER(_lib.ChannelParams)), int(sh.size),
            sh.ctypes.data_as(C.POINTER(C.c_int32)), float(sampling_frequency), C.c_void_p(_ptr(carrier_re)),
            C.c_void_p(_ptr(carrier_im)), C.c_void_p(_ptr(dw_re)), C.c_void_p(_ptr(dw_im)), C.c_void_p(_ptr(accum_re)),
            C.c_void_p(_ptr(accum_im)))
        self.check(rc, "gat_downconvert_and_accumulate")

    def gen_signal(self, re: torch.Tensor, im: torch.Tensor | None, layout: int, num_samples: int,
                   num_ants: int, ant_stride: int, block_stride: int, num_blocks: int, num_channels: int,
                   params_dev: torch.Tensor, sampling_frequency: float, amplitude: float = 1.0,
                   steering_cycles: torch.Tensor | None = None, noise_sigma: float = 0.0, seed: int = 0):
        """``steering_cycles`` (float32 [M] on the device) / ``noise_sigma`` / ``seed``: gat_gen_signal_noisy (per-antenna
        steering phases, complex white Gaussian noise); without them the reference's noise-free generator."""
        if steering_cycles is None and noise_sigma == 0.0:
            rc = self.lib.gat_gen_signal(self._h, C.c_void_p(_ptr(re)), C.c_void_p(_ptr(im)), layout, num_samples,
                                         num_ants, ant_stride, block_stride, num_blocks, num_channels,
                                         C.c_void_p(_ptr(params_dev)), float(sampling_frequency), float(amplitude))
            self.check(rc, "gat_gen_signal")
            return
        if steering_cycles is not None and (steering_cycles.dtype != torch.float32 or steering_cycles.numel() < num_ants):
            raise ValueError("steering_cycles must be float32 [num_ants]")
        rc = self.lib.gat_gen_signal_noisy(self._h, C.c_void_p(_ptr(re)), C.c_void_p(_ptr(im)), layout, num_samples,
                                           num_ants, ant_stride, block_stride, num_blocks, num_channels,
                                           C.c_void_p(_ptr(params_dev)), float(sampling_frequency), float(amplitude),
                                           C.c_void_p(_ptr(steering_cycles)), float(noise_sigma), int(seed) & (2 ** 64 - 1))
        self.check(rc, "gat_gen_signal_noisy")

    def reduce_cplx_multi(self, in_re: torch.Tensor, in_im: torch.Tensor, n: int, cols: int,
                          out_re: torch.Tensor, out_im: torch.Tensor):
        rc = self.lib.gat_reduce_cplx_multi(self._h, C.c_void_p(_ptr(in_re)), C.c_void_p(_ptr(in_im)), n, cols,
                                            C.c_void_p(_ptr(out_re)), C.c_void_p(_ptr(out_im)))
        self.check(rc, "gat_reduce_cplx_multi")

    def open_resident(self, desc: _lib.SignalDesc, num_channels: int, shifts, sampling_frequency: float,
                      buffer_samples: int | None = None, **config):
        """A resident correlator for single-block calls (include/gat.h gat_resident_open): see ``ResidentCorrelator``.
        ``buffer_samples``: samples per antenna of the device buffer ``desc`` points into -- ``correlate`` then refuses a
        ``block_offset`` whose block would end behind it (the kernel reads past its caches: an offset beyond the allocation
        is a GPU page fault, not an exception)."""
        return ResidentCorrelator(self, desc, num_channels, shifts, sampling_frequency, buffer_samples=buffer_samples, **config)

    def params_to_device(self, params: np.ndarray) -> torch.Tensor:
        prm = np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE)
        return torch.from_numpy(prm.view(np.uint8).reshape(-1).copy()).to(self.device)


class ResidentCorrelator:
    """``gat_resident`` (include/gat.h): ONE bounded-lifetime kernel stays on the device for a fixed call geometry; a call
    rings it through pinned host memory (no kernel launch, no stream wait) and returns the correlator outputs as host
    arrays -- the reference's ``@benchmark CUDA.@sync kernel_algorithm(...)`` call (src/benchmarks.jl:120-146) plus the copy
    of its outputs to the host.  ``config``: idle_us, life_ms, max_calls, max_workgroups, host_pollers, doorbell (0 / absent: library defaults).

    The caller makes sure the block's samples are in device memory before ``correlate`` (``torch.cuda.synchronize()`` or
    ``ctx.sync()`` after whatever produced them).  Use as a context manager, or ``close()`` it."""

    def __init__(self, ctx: Context, desc: _lib.SignalDesc, num_channels: int, shifts, sampling_frequency: float,
                 buffer_samples: int | None = None, **config):
        self._limit = None if buffer_samples is None else int(buffer_samples) - int(desc.num_samples)  # largest block offset
        if self._limit is not None and self._limit < 0:
            raise ValueError("buffer_samples is shorter than one block")
        unknown = set(config) - {"idle_us", "life_ms", "max_calls", "max_workgroups", "host_pollers", "doorbell"}
        if unknown:
            raise TypeError(f"unknown resident option(s): {sorted(unknown)}")
        self.ctx = ctx
        self.lib = ctx.lib
        self._desc = desc  # the buffer it describes must outlive the correlator: the caller keeps the tensors
        sh = np.ascontiguousarray(shifts, dtype=np.int32)
        cfg = _lib.ResidentConfig(C.sizeof(_lib.ResidentConfig), *(int(config.get(k, 0)) for k in ("idle_us", "life_ms", "max_calls", "max_workgroups", "host_pollers", "doorbell")))
        self._h = C.c_void_p()
        rc = self.lib.gat_resident_open(ctx._h, C.byref(desc), int(num_channels), int(sh.size), sh.ctypes.data_as(C.POINTER(C.c_int32)),
                                        float(sampling_frequency), C.byref(cfg), C.byref(self._h))
        ctx.check(rc, "gat_resident_open")
        self.shape = (int(num_channels), int(sh.size), int(desc.num_ants))  # [K, L, M] (C order == the ABI's [M x L x K])
        self._re = np.empty(self.shape, dtype=np.float32)
        self._im = np.empty(self.shape, dtype=np.float32)
        self._fn = self.lib.gat_resident_correlate
        self._pre, self._pim = C.c_void_p(self._re.ctypes.data), C.c_void_p(self._im.ctypes.data)
        self._prm_obj, self._prm_ptr = None, None

    def correlate(self, params, block_offset: int = 0):
        """params: structured array of ``num_channels`` records (``_lib.PARAMS_DTYPE``).  Returns (re, im): float32 views
        [K, L, M] that the NEXT call overwrites (copy what has to last)."""
        if params is self._prm_obj:  # a loop that rewrites one record array in place: its pointer is converted once
            pp = self._prm_ptr
        else:
            prm = params if (isinstance(params, np.ndarray) and params.dtype == _lib.PARAMS_DTYPE and params.flags.c_contiguous) \
                else np.ascontiguousarray(params, dtype=_lib.PARAMS_DTYPE)
            if prm.size != self.shape[0]:
                raise ValueError("params must hold num_channels entries")
            pp = prm.ctypes.data_as(C.POINTER(_lib.ChannelParams))
            if prm is params:
                self._prm_obj, self._prm_ptr = params, pp
        if self._limit is not None and not 0 <= block_offset <= self._limit:
            raise ValueError(f"block_offset {block_offset}: the block would lie outside the buffer (0 .. {self._limit})")
        rc = self._fn(self._h, pp, int(block_offset), self._pre, self._pim)
        if rc != 0:
            self.ctx.check(rc, "gat_resident_correlate")
        return self._re, self._im

    def info(self) -> dict:
        i = _lib.ResidentInfo()
        self.ctx.check(self.lib.gat_resident_info_get(self._h, C.byref(i), C.sizeof(i)), "gat_resident_info_get")
        return {n: int(getattr(i, n)) for n, _ in i._fields_}

    def park(self):
        self.ctx.check(self.lib.gat_resident_park(self._h), "gat_resident_park")

    def close(self):
        if self._h and self.ctx._h:  # (a context that was closed first has taken its correlators with it)
            self.lib.gat_resident_close(self._h)
        self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_CONTEXTS: dict = {}


def get_context(device=None, own_stream: bool = False) -> Context:
    """Context for ``device`` bound to PyTorch's CURRENT stream on it (cached) -- or, ``own_stream=True``, the cached
    context with a library-owned stream (the latency regime: see ``Context``)."""
    if not torch.cuda.is_available():
        raise RuntimeError("gpuacceleratedtracking_amd needs a HIP device (no CPU fallback exists)")
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        idx = torch.device(device).index or 0
    if own_stream:
        key = (idx, "own")
        ctx = _CONTEXTS.get(key)
        if ctx is None:
            ctx = Context(idx, "own")
            _CONTEXTS[key] = ctx
        return ctx
    stream = torch.cuda.current_stream(idx)
    key = (idx, stream.cuda_stream)
    ctx = _CONTEXTS.get(key)
    if ctx is None:
        ctx = Context(idx, stream)
        _CONTEXTS[key] = ctx
    return ctx
