"""Shared helpers for the parity tests: build seeded inputs, run the HIP path through the C ABI
(via the Python host layer) and the FP64 oracle on the same inputs, compare."""
import numpy as np

import oracle

# north star: "correlator outputs match the CPU reference within 1e-5 relative on ComplexF32
# accumulators".  Metric: per channel max|dR| / max|R_ref| over its [L, M]
# block, plus element-wise relative error on signal-bearing taps (|R| >= 0.1 max|R|).
RTOL = 1e-5


def system_tables(name):
    lc, fc, _ = oracle.SYSTEMS[name]
    return oracle.codes(name, 32), fc, lc


def make_case(seed, system="GPSL1", N=2500, M=1, L=3, K=1, B=1, fs=None, if_hz=0.0, noise=0.0):
    """Seeded scenario: K channels (distinct PRNs) summed into one antenna signal with
    per-antenna steering phases; per-(block, channel) Doppler / code phase / carrier phase."""
    rng = np.random.default_rng(seed)
    codes, fc, lc = system_tables(system)
    if fs is None:
        fs = N / 1e-3
    prns = rng.permutation(32)[:K]  # every row of both tables is pinned to its ICD (tests/test_oracle_golden.py)
    f = if_hz + rng.uniform(-5e3, 5e3, size=(B, K))
    fcode = fc * (1.0 + (f - if_hz) / 1575.42e6)
    tau = rng.uniform(0, lc, size=(B, K))
    phi = rng.uniform(0, 1, size=(B, K))
    prm = oracle.make_params(np.broadcast_to(prns, (B, K)), fcode, f, tau, phi)
    # signal: sum over channels, built with the oracle's gen_signal (phi in radians there)
    re = np.zeros((M, B * N), dtype=np.float32)
    im = np.zeros((M, B * N), dtype=np.float32)
    steer = np.exp(2j * np.pi * rng.uniform(0, 1, size=M)) if M > 1 else np.ones(1)
    for b in range(B):
        acc = np.zeros(N, dtype=np.complex128)
        for k in range(K):
            r1, i1 = oracle.gen_signal(codes, int(prns[k]), fcode[b, k], fs, f[b, k], tau[b, k],
                                       2 * np.pi * phi[b, k], N, 1)
            acc += r1[0].astype(np.float64) + 1j * i1[0].astype(np.float64)
        if noise > 0:
            acc += noise * (rng.standard_normal(N) + 1j * rng.standard_normal(N))
        x = steer[:, None] * acc[None, :]
        re[:, b * N:(b + 1) * N] = x.real.astype(np.float32)
        im[:, b * N:(b + 1) * N] = x.imag.astype(np.float32)
    shifts = oracle.sample_shifts(L, fs, fc)
    return dict(codes=codes, fc=fc, lc=lc, fs=fs, prm=prm, re=re, im=im, shifts=shifts, N=N, M=M, L=L,
                K=K, B=B, system=system)


def oracle_result(case):
    return oracle.correlate_f64(case["re"], case["im"], case["codes"], case["prm"], case["fs"],
                                case["shifts"], N=case["N"])


def check_close(got, ref, rtol=RTOL, what="", floor_frac=None, abs_floor=0.0):
    """got/ref complex [B, K, L, M].  ``floor_frac`` (default: env GAT_CHECK_FLOOR_FRAC, else 0 = strict): a channel
    whose own largest accumulator is below this fraction of the block's largest one (a short integration whose
    interferers happen to cancel its single tap) is scaled by that floor instead -- a relative error against a
    cancellation residue says nothing about the kernel (scripts/stress_matrix_sweep.py uses 0.01).  ``abs_floor``: the
    same as an absolute magnitude (callers that know the coherent scale N * rms|x| of the integration)."""
    import os
    if floor_frac is None:
        floor_frac = float(os.environ.get("GAT_CHECK_FLOOR_FRAC", "0"))
    got = np.asarray(got, dtype=np.complex128)
    ref = np.asarray(ref, dtype=np.complex128)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    B, K = ref.shape[:2]
    for b in range(B):
        for k in range(K):
            r, g = ref[b, k], got[b, k]
            scale = max(np.abs(r).max(), floor_frac * np.abs(ref[b]).max(), abs_floor)
            e_inf = np.abs(g - r).max() / scale
            assert e_inf <= rtol, f"{what} block {b} chan {k}: norm-wise error {e_inf:.3e} > {rtol}"
            if np.abs(r).max() < scale:  # only with a floor: the whole channel is a cancellation residue,
                continue               # there is no signal-bearing element to judge element-wise
            strong = np.abs(r) >= 0.1 * scale
            rel = (np.abs(g - r)[strong] / np.abs(r)[strong]).max()
            assert rel <= rtol, f"{what} block {b} chan {k}: element-wise error {rel:.3e} > {rtol}"
