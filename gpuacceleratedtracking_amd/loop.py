"""Closed tracking loop around the correlator: the layer Tracking.jl's
``track`` provides and the reference only touches to borrow buffers (``TrackingState``,
src/benchmarks.jl:54-61).  One ``TrackingLoop.step`` = one fused correlate launch for K channels
on one integration block + one ``gat_tracking_update`` launch that turns the accumulators into
the next block's parameters ON THE DEVICE (ping-pong parameter buffers, no host round trip)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .context import Context, get_context
from .gen_signal import make_params
from .signals import GNSSSystem, get_code_frequency
from .tracking import _signal_desc


class TrackingLoop:
    def __init__(self, system: GNSSSystem, prns, num_samples: int, num_ants: int, sampling_frequency: float,
                 correlator_sample_shifts, init_carrier_doppler, init_code_phase, if_hz: float = 0.0,
                 carrier_center_hz: float = 1575.42e6, pll_bandwidth_hz: float = 18.0, dll_bandwidth_hz: float = 1.0,
                 init_carrier_phase=0.0, device=None, ctx: Context | None = None):
        self.ctx = ctx if ctx is not None else get_context(device)
        self.ctx.set_codes(system.codes)
        self.system = system
        prns = np.atleast_1d(np.asarray(prns))
        self.K, self.N, self.M = int(prns.size), int(num_samples), int(num_ants)
        self.fs = float(sampling_frequency)
        self.shifts = np.ascontiguousarray(correlator_sample_shifts, dtype=np.int32)
        self.L = int(self.shifts.size)
        order = np.argsort(self.shifts, kind="stable")
        fc = get_code_frequency(system)
        cfg = _lib.LoopConfig()
        cfg.block_seconds = self.N / self.fs
        cfg.pll_bandwidth_hz, cfg.dll_bandwidth_hz = pll_bandwidth_hz, dll_bandwidth_hz
        cfg.code_freq_nominal_hz, cfg.carrier_center_hz, cfg.if_hz = fc, carrier_center_hz, if_hz
        cfg.code_length, cfg.num_taps = system.code_length, self.L
        cfg.early_index, cfg.late_index = int(order[0]), int(order[-1])
        zero = np.nonzero(self.shifts == 0)[0]
        if zero.size == 0:
            raise ValueError("the tap list needs a prompt tap (shift 0)")
        cfg.prompt_index = int(zero[0])
        cfg.early_late_spacing_chips = float(self.shifts[order[-1]] - self.shifts[order[0]]) * fc / self.fs
        self.config = cfg
        dop = np.broadcast_to(np.asarray(init_carrier_doppler, dtype=np.float64), (self.K,))
        prm = make_params(prns - 1, fc + dop * fc / carrier_center_hz, if_hz + dop, init_code_phase,
                          init_carrier_phase, shape=(1, self.K))
        dev = self.ctx.device
        self._params = [self.ctx.params_to_device(prm), self.ctx.params_to_device(prm)]
        self._cur = 0
        st = np.zeros(self.K, dtype=_lib.LOOP_STATE_DTYPE)
        st["init_carrier_doppler_hz"] = dop
        st["carrier_doppler_hz"] = dop
        self._state = torch.from_numpy(st.view(np.uint8).reshape(-1).copy()).to(dev)
        self.out_re = torch.empty((1, self.K, self.L, self.M), dtype=torch.float32, device=dev)
        self.out_im = torch.empty_like(self.out_re)
        self.blocks_done = 0

    def step(self, re: torch.Tensor, im: torch.Tensor | None = None, start: int = 0):
        """Correlate the block starting at sample ``start`` of the given signal with the current
        parameters, then update them on the device.  Nothing is synchronised."""
        self.ctx.set_codes(self.system.codes)  # the context may have been bound to another system's table meanwhile
        desc = _signal_desc(re, im, self.N, start=start)
        cur, nxt = self._params[self._cur], self._params[1 - self._cur]
        self.ctx.downconvert_and_correlate(desc, cur, 1, self.K, self.shifts, self.fs, self.out_re, self.out_im)
        rc = self.ctx.lib.gat_tracking_update(self.ctx._h, C.c_void_p(self.out_re.data_ptr()),
                                              C.c_void_p(self.out_im.data_ptr()), self.K, self.M,
                                              C.byref(self.config), C.c_void_p(self._state.data_ptr()),
                                              C.c_void_p(cur.data_ptr()), C.c_void_p(nxt.data_ptr()))
        self.ctx.check(rc, "gat_tracking_update")
        self._cur = 1 - self._cur
        self.blocks_done += 1

    def run(self, re: torch.Tensor, im: torch.Tensor | None, num_blocks: int, start: int = 0, keep: bool = True,
            graph: bool = False, out: tuple | None = None):
        """``num_blocks`` consecutive blocks (block b starts at sample start + b * N) through {correlate, update}
        in one native call (``gat_tracking_run``): the same launches as ``num_blocks`` x ``step`` without a
        host round trip per block.  Returns the accumulators of every block as device tensors (re, im)
        [num_blocks, K, L, M] when ``keep`` (into ``out`` = (re, im) if given), else only the last block's in
        ``out_re`` / ``out_im``.  ``graph``: GAT_FLAG_GRAPH -- calls that repeat with the same buffers replay an
        instantiated hipGraph (the context must be bound to a non-default stream)."""
        nb = int(num_blocks)
        self.ctx.set_codes(self.system.codes)  # the context may have been bound to another system's table meanwhile
        desc = _signal_desc(re, im, self.N, start=start)
        ntot = re.shape[-2] if im is None else re.shape[-1]
        if start + nb * self.N > ntot:
            raise ValueError("signal shorter than start + num_blocks * num_samples")
        if keep:
            if out is not None:
                acc_re, acc_im = out
            else:
                acc_re = torch.empty((nb, self.K, self.L, self.M), dtype=torch.float32, device=self.ctx.device)
                acc_im = torch.empty_like(acc_re)
            stride = self.K * self.L * self.M
        else:
            acc_re, acc_im, stride = self.out_re, self.out_im, 0
        a, b = self._params[self._cur], self._params[1 - self._cur]
        is_b = C.c_int32(0)
        rc = self.ctx.lib.gat_tracking_run(self.ctx._h, C.byref(desc), nb, self.K, self.L,
                                           self.shifts.ctypes.data_as(C.POINTER(C.c_int32)), self.fs,
                                           C.byref(self.config), C.c_void_p(self._state.data_ptr()),
                                           C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                           C.c_void_p(acc_re.data_ptr()), C.c_void_p(acc_im.data_ptr()), stride,
                                           _lib.GAT_FLAG_GRAPH if graph else 0,
                                           C.byref(is_b))
        self.ctx.check(rc, "gat_tracking_run")
        if is_b.value:
            self._cur = 1 - self._cur
        self.blocks_done += nb
        if keep:
            self.out_re.copy_(acc_re[-1:])
            self.out_im.copy_(acc_im[-1:])
        return acc_re, acc_im

    def params(self) -> np.ndarray:
        """Current per-channel parameters (host copy; synchronises)."""
        return self._params[self._cur].cpu().numpy().view(_lib.PARAMS_DTYPE).copy()

    def state(self) -> np.ndarray:
        return self._state.cpu().numpy().view(_lib.LOOP_STATE_DTYPE).copy()

    def accumulators(self) -> np.ndarray:
        """complex64 [K, L, M] of the last block."""
        return (self.out_re.cpu().numpy() + 1j * self.out_im.cpu().numpy())[0].astype(np.complex64)


class ResidentTrackingLoop:
    """The closed loop with the HOST in it, as in the reference's receiver (Tracking.jl's discriminators and loop filters run on
    the CPU): one step = one call rung into a resident correlator (``gat_resident_correlate``: no kernel launch, outputs
    arrive on the host) + ``gat_tracking_update_host`` (the arithmetic of the device's ``gat_tracking_update``, csrc/gat_loop.h)
    that turns the accumulators into the next block's parameters.  Same constructor as ``TrackingLoop`` plus the signal
    buffer the blocks live in; up to 16 channels.  ``run`` is the native loop (gat_resident_tracking_run)."""

    def __init__(self, system: GNSSSystem, prns, num_samples: int, num_ants: int, sampling_frequency: float,
                 correlator_sample_shifts, init_carrier_doppler, init_code_phase, re: torch.Tensor, im: torch.Tensor | None = None,
                 if_hz: float = 0.0, carrier_center_hz: float = 1575.42e6, pll_bandwidth_hz: float = 18.0,
                 dll_bandwidth_hz: float = 1.0, init_carrier_phase=0.0, device=None, ctx: Context | None = None, **resident_config):
        # the loop configuration, parameters and state: TrackingLoop's, kept on the host
        dev_loop = TrackingLoop(system, prns, num_samples, num_ants, sampling_frequency, correlator_sample_shifts, init_carrier_doppler,
                                init_code_phase, if_hz, carrier_center_hz, pll_bandwidth_hz, dll_bandwidth_hz, init_carrier_phase, device, ctx)
        self.ctx, self.system, self.config = dev_loop.ctx, system, dev_loop.config
        self.K, self.N, self.M, self.L, self.fs, self.shifts = dev_loop.K, dev_loop.N, dev_loop.M, dev_loop.L, dev_loop.fs, dev_loop.shifts
        self._cur = dev_loop.params().reshape(-1).copy()
        self._next = self._cur.copy()
        self._state = dev_loop.state().copy()
        self._signal = (re, im)  # kept alive: the resident kernel reads it
        desc = _signal_desc(re, im, self.N, start=0)
        self._samples = int(re.shape[-2] if im is None else re.shape[-1])  # per antenna: every block has to end inside
        torch.cuda.current_stream(self.ctx.device).synchronize()  # the signal is on the device before the first ring
        self.resident = self.ctx.open_resident(desc, self.K, self.shifts, self.fs, buffer_samples=self._samples, **resident_config)
        self.resident_workgroups = self.resident.info()["workgroups"]
        self._lib = self.ctx.lib
        self._fn = self._lib.gat_tracking_update_host
        self.acc_re = self.acc_im = None
        self.blocks_done = 0

    def step(self, start: int = 0):
        """Correlate the block that starts ``start`` samples into the buffer with the current parameters, update them."""
        re, im = self.resident.correlate(self._cur, block_offset=start)
        rc = self._fn(C.c_void_p(re.ctypes.data), C.c_void_p(im.ctypes.data), self.K, self.M, C.byref(self.config),
                      C.c_void_p(self._state.ctypes.data), C.c_void_p(self._cur.ctypes.data), C.c_void_p(self._next.ctypes.data))
        if rc != 0:
            raise _lib.GatError(rc, "gat_tracking_update_host")
        self._cur, self._next = self._next, self._cur
        self.acc_re, self.acc_im = re, im
        self.blocks_done += 1

    def run(self, num_blocks: int, start: int = 0, block_stride: int | None = None, keep_all: bool = False):
        """``num_blocks`` consecutive blocks from native code (``gat_resident_tracking_run``: {resident call, host update} per
        block without a trip through Python).  The blocks are ``block_stride`` samples apart (default: back to back) and all
        of them are on the device already.  ``keep_all``: returns complex64 [num_blocks, K, L, M] of every block's
        accumulators; otherwise only the last block's are kept (``accumulators()``)."""
        nb = int(num_blocks)
        if nb <= 0:
            return None
        n = self.K * self.L * self.M
        re = np.empty((nb if keep_all else 1, self.K, self.L, self.M), np.float32)
        im = np.empty_like(re)
        stride = self.N if block_stride is None else int(block_stride)
        # (the resident kernel reads the blocks with system-scope loads: a block outside the allocation is a GPU page fault)
        if start < 0 or stride < 0 or int(start) + (nb - 1) * stride + self.N > self._samples:
            raise ValueError("signal shorter than start + (num_blocks - 1) * block_stride + num_samples")
        rc = self._lib.gat_resident_tracking_run(self.resident._h, nb, int(start), stride, C.byref(self.config), C.c_void_p(self._state.ctypes.data),
                                                 C.c_void_p(self._cur.ctypes.data), C.c_void_p(re.ctypes.data), C.c_void_p(im.ctypes.data), n if keep_all else 0)
        self.ctx.check(rc, "gat_resident_tracking_run")
        self.acc_re, self.acc_im = re[-1], im[-1]
        self.blocks_done += nb
        return (re + 1j * im).astype(np.complex64) if keep_all else None

    def params(self) -> np.ndarray:
        return self._cur.copy()

    def state(self) -> np.ndarray:
        return self._state.copy()

    def accumulators(self) -> np.ndarray:
        """complex64 [K, L, M] of the last block."""
        return (self.acc_re + 1j * self.acc_im).astype(np.complex64)

    def close(self):
        self.resident.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
