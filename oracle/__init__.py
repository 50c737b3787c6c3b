"""CPU oracle for the downconvert + correlate path -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/gat_oracle.c`` (the C restatement of the reference's algorithm;
every function there cites the reference file:line it follows) plus a second, independent numpy
restatement (``np_*`` functions) used to cross-check the C one.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``gpuacceleratedtracking_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIB_NATIVE = None


class Params(C.Structure):
    """Mirror of ``gat_oracle_params`` (one satellite channel in one integration block)."""

    _fields_ = [
        ("prn0", C.c_int32),
        ("pad_", C.c_int32),
        ("code_freq_hz", C.c_double),
        ("carrier_freq_hz", C.c_double),
        ("code_phase_chips", C.c_double),
        ("carrier_phase_cycles", C.c_double),
    ]


PARAMS_DTYPE = np.dtype(
    [
        ("prn0", "<i4"),
        ("pad_", "<i4"),
        ("code_freq_hz", "<f8"),
        ("carrier_freq_hz", "<f8"),
        ("code_phase_chips", "<f8"),
        ("carrier_phase_cycles", "<f8"),
    ]
)


def build(native: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile).  Returns the .so path."""
    target = ["native"] if native else []
    subprocess.run(["make", "-s", "-C", _HERE] + target, check=True)
    return os.path.join(_HERE, "libgat_oracle_native.so" if native else "libgat_oracle.so")


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def lib(native: bool = False):
    """Load (building if necessary) the oracle shared library."""
    global _LIB, _LIB_NATIVE
    if native:
        if _LIB_NATIVE is None:
            try:
                _LIB_NATIVE = _bind(C.CDLL(build(native=True)))
            except Exception:  # pragma: no cover - fall back to the portable build
                _LIB_NATIVE = False
        if _LIB_NATIVE:
            return _LIB_NATIVE
    if _LIB is None:
        path = os.path.join(_HERE, "libgat_oracle.so")
        src = os.path.join(_HERE, "gat_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = _bind(C.CDLL(path))
    return _LIB


def _bind(l):
    i8p, i32p, fp, dp = (C.POINTER(t) for t in (C.c_int8, C.c_int32, C.c_float, C.c_double))
    l.gat_oracle_code_gpsl1.argtypes = [C.c_int, i8p]
    l.gat_oracle_code_gpsl5.argtypes = [C.c_int, i8p]
    l.gat_oracle_sample_shifts.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, i32p]
    l.gat_oracle_gen_signal.argtypes = [i8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_double, C.c_int64, C.c_int, C.c_int64, fp, fp]
    l.gat_oracle_gen_code_replica.argtypes = [i8p, C.c_int, C.c_int, C.c_double, C.c_double,
                                              C.c_double, C.c_int64, C.c_int64, fp]
    l.gat_oracle_correlate_f64.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int, i8p, C.c_int,
                                           C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_int, i32p, dp, dp]
    l.gat_oracle_dc_f32_4pass.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int, i8p, C.c_int,
                                          C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                          C.c_double, C.c_int, i32p, fp, fp, fp]
    pp = C.POINTER(Params)
    l.gat_oracle_dc_f32_batched.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int64, C.c_int,
                                            C.c_int, C.c_int, i8p, C.c_int, pp, C.c_double, C.c_int,
                                            i32p, C.c_int, fp, fp]
    l.gat_oracle_correlate_f64_batched.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int64,
                                                   C.c_int64, C.c_int, C.c_int, C.c_int, i8p,
                                                   C.c_int, pp, C.c_double, C.c_int, i32p, dp, dp]
    l.gat_oracle_reduce_cplx_multi.argtypes = [fp, fp, C.c_int64, C.c_int, dp, dp]
    l.gat_oracle_dc_f32_time.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int, i8p, C.c_int, C.c_int, C.c_double,
                                         C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, i32p, C.c_int, dp, fp, fp]
    l.gat_oracle_dc_f32_profile.argtypes = [fp, fp, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, i8p,
                                            C.c_int, pp, C.c_double, C.c_int, i32p, fp, fp, dp]
    return l


# ------------------------------------------------------------------------------------------
# Code tables
# ------------------------------------------------------------------------------------------
SYSTEMS = {
    # name: (code length, code frequency Hz, generator symbol)
    "GPSL1": (1023, 1.023e6, "gat_oracle_code_gpsl1"),
    "GPSL5": (10230, 10.23e6, "gat_oracle_code_gpsl5"),
}


def codes(system: str, num_prns: int = 32) -> np.ndarray:
    """int8 +-1 table, C-order [num_prns, Lc] == column-major [Lc x P] of the reference
    (``codes[chip, prn]``, src/algorithms.jl:185)."""
    lc, _, sym = SYSTEMS[system]
    out = np.empty((num_prns, lc), dtype=np.int8)
    fn = getattr(lib(), sym)
    for p in range(num_prns):
        row = np.empty(lc, dtype=np.int8)
        rc = fn(p + 1, _p(row, C.c_int8))
        assert rc == 0
        out[p] = row
    return out


def sample_shifts(L: int, fs: float, fc: float, spacing: float = 0.5) -> np.ndarray:
    out = np.empty(L, dtype=np.int32)
    assert lib().gat_oracle_sample_shifts(L, fs, fc, spacing, _p(out, C.c_int32)) == 0
    return out


# ------------------------------------------------------------------------------------------
# Signal synthesis / replica / correlator
# ------------------------------------------------------------------------------------------
def gen_signal(codes_tbl, prn0, fc, fs, f, tau, phi_rad, N, M):
    """Planar (re, im), each float32 array of shape [M, N] (C-order == column-major [N x M])."""
    re = np.empty((M, N), dtype=np.float32)
    im = np.empty((M, N), dtype=np.float32)
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    rc = lib().gat_oracle_gen_signal(_p(ct, C.c_int8), ct.shape[1], prn0, fc, fs, f, tau, phi_rad,
                                     N, M, N, _p(re, C.c_float), _p(im, C.c_float))
    assert rc == 0
    return re, im


def gen_code_replica(codes_tbl, prn0, fc, fs, tau, first_shift, count):
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    rep = np.empty(count, dtype=np.float32)
    rc = lib().gat_oracle_gen_code_replica(_p(ct, C.c_int8), ct.shape[1], prn0, fc, fs, tau,
                                           first_shift, count, _p(rep, C.c_float))
    assert rc == 0
    return rep


def make_params(prn0, fc, f, tau, phi_cycles, shape=None) -> np.ndarray:
    """Structured array of per-(block, channel) parameters, broadcasting scalars."""
    arrs = np.broadcast_arrays(np.asarray(prn0), np.asarray(fc, dtype=np.float64),
                               np.asarray(f, dtype=np.float64), np.asarray(tau, dtype=np.float64),
                               np.asarray(phi_cycles, dtype=np.float64))
    if shape is not None:
        arrs = [np.broadcast_to(a, shape) for a in arrs]
    out = np.zeros(arrs[0].shape, dtype=PARAMS_DTYPE)
    out["prn0"], out["code_freq_hz"], out["carrier_freq_hz"] = arrs[0], arrs[1], arrs[2]
    out["code_phase_chips"], out["carrier_phase_cycles"] = arrs[3], arrs[4]
    return out


def correlate_f64(re, im, codes_tbl, params, fs, shifts, N=None, blk_stride=None, chan_stride=0):
    """FP64 oracle.  re/im: float32 [M, ld] planar (or [K, M, ld] when chan_stride != 0).
    params: structured [B, K].  Returns complex128 [B, K, L, M]."""
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    params = np.ascontiguousarray(params)
    B, K = params.shape
    M = re.shape[-2]
    ld = re.shape[-1]
    if N is None:
        N = ld // B
    if blk_stride is None:
        blk_stride = N
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    L = sh.size
    o_re = np.empty((B, K, L, M), dtype=np.float64)
    o_im = np.empty_like(o_re)
    rc = lib().gat_oracle_correlate_f64_batched(
        _p(re, C.c_float), _p(im, C.c_float), ld, blk_stride, chan_stride, N, M, B, K,
        _p(ct, C.c_int8), ct.shape[1], params.ctypes.data_as(C.POINTER(Params)), fs, L,
        _p(sh, C.c_int32), _p(o_re, C.c_double), _p(o_im, C.c_double))
    assert rc == 0
    return o_re + 1j * o_im


def dc_f32(re, im, codes_tbl, params, fs, shifts, N=None, blk_stride=None, threads=1, native=False):
    """FP32 4-pass CPU baseline ("port" of the Tracking.jl CPU structure).
    Returns complex64 [B, K, L, M]."""
    l = lib(native=native)
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    params = np.ascontiguousarray(params)
    B, K = params.shape
    M, ld = re.shape
    if N is None:
        N = ld // B
    if blk_stride is None:
        blk_stride = N
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    L = sh.size
    o_re = np.empty((B, K, L, M), dtype=np.float32)
    o_im = np.empty_like(o_re)
    rc = l.gat_oracle_dc_f32_batched(
        _p(re, C.c_float), _p(im, C.c_float), ld, blk_stride, N, M, B, K, _p(ct, C.c_int8),
        ct.shape[1], params.ctypes.data_as(C.POINTER(Params)), fs, L, _p(sh, C.c_int32), threads,
        _p(o_re, C.c_float), _p(o_im, C.c_float))
    assert rc == 0
    return o_re + 1j * o_im


def dc_f32_profile(re, im, codes_tbl, params, fs, shifts, N=None, blk_stride=None, native=False):
    """The 4-pass CPU baseline on ONE thread with the wall time of each pass accumulated over all (block, channel)
    calls.  Returns (complex64 [B, K, L, M], seconds [4]: code replica, carrier replica, downconvert, correlate)."""
    l = lib(native=native)
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    params = np.ascontiguousarray(params)
    B, K = params.shape
    M, ld = re.shape
    if N is None:
        N = ld // B
    if blk_stride is None:
        blk_stride = N
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    L = sh.size
    o_re = np.empty((B, K, L, M), dtype=np.float32)
    o_im = np.empty_like(o_re)
    secs = np.zeros(4, dtype=np.float64)
    rc = l.gat_oracle_dc_f32_profile(
        _p(re, C.c_float), _p(im, C.c_float), ld, blk_stride, N, M, B, K, _p(ct, C.c_int8), ct.shape[1],
        params.ctypes.data_as(C.POINTER(Params)), fs, L, _p(sh, C.c_int32), _p(o_re, C.c_float), _p(o_im, C.c_float),
        _p(secs, C.c_double))
    assert rc == 0
    return o_re + 1j * o_im, secs


def dc_f32_time(re, im, codes_tbl, prn0, fc, fs, f, tau, phi_cycles, shifts, reps, native=True):
    """``reps`` single-block calls of the 4-pass CPU baseline on ONE thread, each timed inside C (the reference's
    ``@benchmark Tracking.downconvert_and_correlate!(...)``, src/benchmarks.jl:63-79).  re/im float32 [M, N].
    Returns (times_ns float64 [reps], complex64 [L, M])."""
    l = lib(native=native)
    re = np.ascontiguousarray(re, dtype=np.float32)
    im = np.ascontiguousarray(im, dtype=np.float32)
    ct = np.ascontiguousarray(codes_tbl, dtype=np.int8)
    M, N = re.shape
    sh = np.ascontiguousarray(shifts, dtype=np.int32)
    times = np.zeros(reps, dtype=np.float64)
    o_re = np.empty((sh.size, M), dtype=np.float32)
    o_im = np.empty_like(o_re)
    rc = l.gat_oracle_dc_f32_time(_p(re, C.c_float), _p(im, C.c_float), N, N, M, _p(ct, C.c_int8), ct.shape[1], prn0, fc,
                                  fs, f, tau, phi_cycles, sh.size, _p(sh, C.c_int32), reps, _p(times, C.c_double),
                                  _p(o_re, C.c_float), _p(o_im, C.c_float))
    assert rc == 0
    return times, o_re + 1j * o_im


def reduce_cplx_multi(in_re, in_im):
    """Column sums of a planar complex [ML, n] (C-order) array -> complex128 [ML]."""
    in_re = np.ascontiguousarray(in_re, dtype=np.float32)
    in_im = np.ascontiguousarray(in_im, dtype=np.float32)
    ml, n = in_re.shape
    o_re = np.empty(ml, dtype=np.float64)
    o_im = np.empty(ml, dtype=np.float64)
    rc = lib().gat_oracle_reduce_cplx_multi(_p(in_re, C.c_float), _p(in_im, C.c_float), n, ml,
                                            _p(o_re, C.c_double), _p(o_im, C.c_double))
    assert rc == 0
    return o_re + 1j * o_im


# ------------------------------------------------------------------------------------------
# Independent numpy restatement (cross-check of the C oracle; small cases)
# ------------------------------------------------------------------------------------------
def np_code_gpsl1(prn: int) -> np.ndarray:
    """C/A code by the G2-delay formulation (IS-GPS-200 Table 3-Ia 'code delay chips'),
    deliberately different from the tap-selector formulation in the C file."""
    delays = [5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471,
              472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862]

    def lfsr(taps):
        reg = [1] * 10
        out = np.empty(1023, dtype=np.int64)
        for i in range(1023):
            out[i] = reg[9]
            fb = 0
            for t in taps:
                fb ^= reg[t - 1]
            reg = [fb] + reg[:9]
        return out

    g1 = lfsr([3, 10])
    g2 = lfsr([2, 3, 6, 8, 9, 10])
    g2d = np.roll(g2, delays[prn - 1])
    return (1 - 2 * (g1 ^ g2d)).astype(np.int8)


def np_correlate(re, im, code_row, fc, fs, f, tau, phi_cycles, shifts):
    """Direct numpy evaluation of SURVEY section 0 / src/algorithms.jl:170-187.
    re/im [M, N] float32 -> complex128 [L, M]."""
    M, N = re.shape
    n = np.arange(N, dtype=np.int64)
    ratio = np.float64(fc) / np.float64(fs)
    th = 2.0 * np.pi * (n.astype(np.float64) * f / fs + phi_cycles)
    car = np.cos(th) - 1j * np.sin(th)
    x = re.astype(np.float64) + 1j * im.astype(np.float64)
    dw = x * car[None, :]
    out = np.empty((len(shifts), M), dtype=np.complex128)
    lc = code_row.size
    for li, s in enumerate(shifts):
        p = ratio * (n + int(s)).astype(np.float64) + np.float64(tau)
        idx = np.mod(np.floor(p).astype(np.int64), lc)
        out[li] = (dw * code_row[idx].astype(np.float64)[None, :]).sum(axis=1)
    return out


def np_tracking_update(acc, cfg: dict, state: dict, cur: np.ndarray):
    """Numpy restatement of gat_tracking_update (include/gat.h): textbook Costas-PLL (3rd-order
    bilinear filter, Kaplan & Hegarty table 5.6) + carrier-aided normalised early-minus-late DLL
    (2nd-order bilinear).  No reference implementation exists in the reference tree (Tracking.jl is
    un-vendored): UNPINNED -- it pins the device kernel to these equations only.
    acc complex [K, L, M]; cfg: dict of gat_loop_config fields; state: dict of float64 [K] arrays
    (gat_loop_state fields); cur: structured params [K].  Returns (next_params, new_state)."""
    T = cfg["block_seconds"]
    P = acc[:, cfg["prompt_index"], :].astype(np.complex128).sum(axis=1)
    E = acc[:, cfg["early_index"], :].astype(np.complex128).sum(axis=1)
    Lt = acc[:, cfg["late_index"], :].astype(np.complex128).sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        pll = np.where(P == 0, 0.0, np.arctan(P.imag / P.real) / (2 * np.pi))
    e, l = np.abs(E), np.abs(Lt)
    dll = np.where(e + l > 0, 0.5 * (2.0 - cfg["early_late_spacing_chips"]) * (l - e) / np.where(e + l > 0, e + l, 1), 0.0)
    st = {k: np.array(v, dtype=np.float64, copy=True) for k, v in state.items()}
    w0p = cfg["pll_bandwidth_hz"] / 0.7845
    in1 = w0p ** 3 * pll
    out1 = st["pll_acc1"] + 0.5 * T * in1
    st["pll_acc1"] += T * in1
    in2 = out1 + 1.1 * w0p ** 2 * pll
    out2 = st["pll_acc2"] + 0.5 * T * in2
    st["pll_acc2"] += T * in2
    carrier_rate = out2 + 2.4 * w0p * pll
    w0d = cfg["dll_bandwidth_hz"] / 0.53
    ind = w0d ** 2 * dll
    outd = st["dll_acc"] + 0.5 * T * ind
    st["dll_acc"] += T * ind
    code_rate = outd + 1.414 * w0d * dll
    car_dop = st["init_carrier_doppler_hz"] + carrier_rate
    code_dop = code_rate + car_dop * cfg["code_freq_nominal_hz"] / cfg["carrier_center_hz"]
    nxt = cur.copy()
    phi = cur["carrier_phase_cycles"] + cur["carrier_freq_hz"] * T
    nxt["carrier_phase_cycles"] = phi - np.floor(phi)
    tau = cur["code_phase_chips"] + cur["code_freq_hz"] * T
    nxt["code_phase_chips"] = tau - np.floor(tau / cfg["code_length"]) * cfg["code_length"]
    nxt["carrier_freq_hz"] = cfg["if_hz"] + car_dop
    nxt["code_freq_hz"] = cfg["code_freq_nominal_hz"] + code_dop
    st["carrier_doppler_hz"], st["code_doppler_hz"] = car_dop, code_dop
    st["last_pll_error_cycles"], st["last_dll_error_chips"] = pll, dll
    st["prompt_power"] = np.abs(P) ** 2
    return nxt, st
