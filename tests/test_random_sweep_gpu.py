"""Seeded random sweep on the GPU: 60 random shapes / layouts / parameters through the C ABI against
the FP64 oracle (north-star tolerance 1e-5), covering whatever the hand-written grids miss: odd
antenna counts, tap counts and orders, ragged N, IF in the MHz range, all four sample formats, both
second-stage flavours, host- and device-resident parameters."""
import numpy as np
import pytest

from tests.helpers import check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def _random_config(rng):
    system = rng.choice(["GPSL1", "GPSL1", "GPSL5"])
    M = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32]))
    L = int(rng.choice([1, 2, 3, 3, 3, 5, 7, 9]))
    K = int(rng.choice([1, 1, 2, 3, 6]))
    B = int(rng.choice([1, 1, 2, 5]))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 5000), rng.integers(5000, 30000)]))
    if rng.random() < 0.6:
        N = max(8, N - N % 8)  # aligned -> vector kernels (and MFMA where eligible)
    fs = float(rng.choice([2.048e6, 4e6, 10e6, 25e6]))
    if_hz = float(rng.choice([0.0, 1.0e5, 0.23 * fs]))
    layout = int(rng.integers(0, 4))
    return dict(system=system, N=N, M=M, L=L, K=K, B=B, fs=fs, if_hz=if_hz), layout


@pytest.mark.parametrize("seed", range(60))
def test_random_case(g, seed):
    import torch
    rng = np.random.default_rng(20240 + seed)
    cfg, layout = _random_config(rng)
    case = make_case(9000 + seed, **cfg)
    if rng.random() < 0.5:  # unsorted / irregular taps
        span = max(2, int(2 * cfg["fs"] / 1.023e6))
        case["shifts"] = rng.integers(-span, span + 1, size=cfg["L"]).astype(np.int32)
    flags = g.GAT_FLAG_ATOMIC if rng.random() < 0.25 else 0
    dev = g.get_context().device
    re, im = case["re"], case["im"]
    if layout >= 2:  # integer ingest: quantise, the oracle sees the exact integers
        amp, dt = ((2000.0, np.int16), (30.0, np.int8))[layout - 2]
        lim = np.iinfo(dt)
        qre = np.clip(np.rint(re * amp / cfg["K"]), lim.min, lim.max).astype(dt)
        qim = np.clip(np.rint(im * amp / cfg["K"]), lim.min, lim.max).astype(dt)
        case["re"], case["im"] = qre.astype(np.float32), qim.astype(np.float32)
        x = torch.from_numpy(np.stack([qre, qim], axis=-1)).to(dev)
        sig = (x, None)
    elif layout == 1:
        sig = (torch.from_numpy(np.stack([re, im], axis=-1)).to(dev), None)
    else:
        sig = (torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev))
    ref = oracle_result(case)
    sysobj = g.GNSSDICT[cfg["system"]](use_gpu=True)
    op = g.StreamCorrelator(sysobj, cfg["N"], cfg["M"], cfg["B"], cfg["K"], case["shifts"], cfg["fs"], flags=flags)
    p = case["prm"]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    if rng.random() < 0.5:
        op.set_params(prm)
        op(*sig)
    else:  # host-parameter entry point (gat_downconvert_and_correlate)
        op.ctx.downconvert_and_correlate(op.describe(*sig), prm, cfg["B"], cfg["K"], case["shifts"], cfg["fs"],
                                         op.out_re, op.out_im, flags)
    import os
    frac = float(os.environ.get("GAT_CHECK_FLOOR_FRAC", "0"))  # stress runs only, see test_random_matrix_case
    abs_floor = frac * cfg["N"] * float(np.sqrt(np.mean(case["re"].astype(np.float64) ** 2 + case["im"].astype(np.float64) ** 2)))
    check_close(op.result(), ref, abs_floor=abs_floor,
                what=f"seed {seed}: {cfg} layout {layout} shifts {case['shifts'].tolist()} flags {flags}")


def _random_matrix_config(rng):
    """Antenna-rich shapes (M % 16 == 0): the planner sends most of these to the matrix-core kernels."""
    system = rng.choice(["GPSL1", "GPSL1", "GPSL5"])
    M = int(rng.choice([16, 16, 32, 48, 64]))
    L = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 6]))
    K = int(rng.choice([2, 3, 4, 5, 6, 9, 13, 21, 30]))
    B = int(rng.choice([1, 1, 2, 3]))
    N = int(rng.choice([rng.integers(8, 200), rng.integers(200, 3000), rng.integers(3000, 40000)]))
    N = max(8, N - N % 8)
    fs = float(rng.choice([4e6, 10e6, 25e6, 50e6]))
    if_hz = float(rng.choice([0.0, 2.5e5, 0.2 * fs]))
    layout = int(rng.integers(0, 4))
    return dict(system=system, N=N, M=M, L=L, K=K, B=B, fs=fs, if_hz=if_hz), layout


@pytest.mark.parametrize("seed", range(40))
def test_random_matrix_case(g, seed):
    """The same check on shapes the matrix-core kernels take, in every kernel-selection mode the shape allows
    (auto / f32 MFMA / vector): all must match the oracle, whichever kernel the planner picked."""
    import torch
    rng = np.random.default_rng(77000 + seed)
    cfg, layout = _random_matrix_config(rng)
    case = make_case(5000 + seed, **cfg)
    if rng.random() < 0.4:  # unsorted / irregular taps
        span = max(2, int(1.5 * cfg["fs"] / 1.023e6))
        case["shifts"] = rng.integers(-span, span + 1, size=cfg["L"]).astype(np.int32)
    flags = g.GAT_FLAG_ATOMIC if rng.random() < 0.2 else 0
    ctx = g.get_context()
    dev = ctx.device
    re, im = case["re"], case["im"]
    if layout >= 2:
        amp, dt = ((2000.0, np.int16), (30.0, np.int8))[layout - 2]
        lim = np.iinfo(dt)
        qre = np.clip(np.rint(re * amp / cfg["K"]), lim.min, lim.max).astype(dt)
        qim = np.clip(np.rint(im * amp / cfg["K"]), lim.min, lim.max).astype(dt)
        case["re"], case["im"] = qre.astype(np.float32), qim.astype(np.float32)
        sig = (torch.from_numpy(np.stack([qre, qim], axis=-1)).to(dev), None)
    elif layout == 1:
        sig = (torch.from_numpy(np.stack([re, im], axis=-1)).to(dev), None)
    else:
        sig = (torch.from_numpy(re).to(dev), torch.from_numpy(im).to(dev))
    ref = oracle_result(case)
    sysobj = g.GNSSDICT[cfg["system"]](use_gpu=True)
    p = case["prm"]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    # stress runs only (GAT_CHECK_FLOOR_FRAC > 0): residues of cancellation are judged against the coherent scale
    import os
    frac = float(os.environ.get("GAT_CHECK_FLOOR_FRAC", "0"))
    abs_floor = frac * cfg["N"] * float(np.sqrt(np.mean(case["re"].astype(np.float64) ** 2 + case["im"].astype(np.float64) ** 2)))
    kinds = []
    try:
        for mode in (g.GAT_MC_AUTO, g.GAT_MC_F32, g.GAT_MC_VECTOR, g.GAT_MC_BF16_SPLIT):
            ctx.set_matrix_core(mode)
            op = g.StreamCorrelator(sysobj, cfg["N"], cfg["M"], cfg["B"], cfg["K"], case["shifts"], cfg["fs"], flags=flags)
            op.set_params(prm)
            op(*sig)
            kinds.append(ctx.last_launch_info()["matrix_core"])
            check_close(op.result(), ref, abs_floor=abs_floor,
                        what=f"seed {seed} mode {mode} kernel {kinds[-1]}: {cfg} layout {layout} shifts {case['shifts'].tolist()} flags {flags}")
    finally:
        ctx.set_matrix_core(1)
    assert kinds[2] == 0 and kinds[1] in (0, 1) and kinds[0] in (0, 1, 2) and kinds[3] in (0, 1, 2)
