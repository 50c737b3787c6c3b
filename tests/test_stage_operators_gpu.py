"""The stand-alone stages of the reference's algorithm ladder that the fused correlator makes unnecessary, kept as
operators so that the reference's own tests of them have a counterpart (`_nsat_`
replica of row A3).  Run with -m gpu."""
import numpy as np
import pytest

import oracle
from tests.helpers import RTOL, check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as gat
    gat.load_library()
    return gat


# ---- A8: cpu_reduce_partial_sum / cuda_reduce_partial_sum (src/algorithms.jl:1-11) --------------------------------
@pytest.mark.parametrize("blocks,M,L", [(10, 4, 3), (625, 4, 3), (77, 1, 7), (1300, 16, 3)])
def test_partial_sum_second_stage(g, blocks, M, L):
    """Algorithm 1's structure (src/algorithms.jl:896-923): per-block partial sums [blocks x M x L], then a second
    stage over the blocks -- on the host (cpu_reduce_partial_sum) or on the device (cuda_reduce_partial_sum).  The
    partials here are REAL ones: one coherent integration cut into `blocks` pieces, each correlated by the fused kernel
    with continuing code / carrier phase; the second stage must reproduce the oracle's single long integration."""
    import torch
    n = 256  # samples per piece
    N = blocks * n
    case = make_case(4242 + blocks, system="GPSL1", N=N, M=M, L=L, K=1, B=1, fs=4e6)
    ref = oracle_result(case)[0, 0]  # [L, M]
    p = case["prm"][0, 0]
    b = np.arange(blocks, dtype=np.float64)
    ratio, step = p["code_freq_hz"] / case["fs"], p["carrier_freq_hz"] / case["fs"]
    prm = g.make_params(int(p["prn0"]), p["code_freq_hz"], p["carrier_freq_hz"],
                        (p["code_phase_chips"] + ratio * n * b)[:, None], (p["carrier_phase_cycles"] + step * n * b)[:, None],
                        shape=(blocks, 1))
    sysobj = g.GPSL1(use_gpu=True)
    ctx = g.get_context()
    op = g.StreamCorrelator(sysobj, n, M, blocks, 1, case["shifts"], case["fs"], ctx=ctx)
    op.set_params(prm)
    op(torch.from_numpy(case["re"]).to(ctx.device), torch.from_numpy(case["im"]).to(ctx.device))
    # [B, 1, L, M] -> the reference's partial-sum layout [blocks x M x L] == C-order [L, M, blocks]
    part_re = op.out_re[:, 0].permute(1, 2, 0).contiguous()
    part_im = op.out_im[:, 0].permute(1, 2, 0).contiguous()
    host = g.cpu_reduce_partial_sum(part_re, part_im)               # numpy complex [L, M]
    dev_re, dev_im = g.cuda_reduce_partial_sum(part_re, part_im)   # torch [L, M]
    dev = dev_re.cpu().numpy() + 1j * dev_im.cpu().numpy()
    for name, got in (("cpu_reduce_partial_sum", host), ("cuda_reduce_partial_sum", dev)):
        check_close(got[None, None], ref[None, None], what=f"{name} blocks={blocks}")
    # the oracle's column sum of the same partials (what `sum(partial, dims=1)` is in the reference's tests)
    colsum = oracle.reduce_cplx_multi(part_re.cpu().numpy().reshape(L * M, blocks), part_im.cpu().numpy().reshape(L * M, blocks))
    assert np.abs(dev.reshape(-1) - colsum).max() <= 1e-6 * np.abs(colsum).max()


# ---- A4: downconvert_and_accumulate_strided_kernel! (src/algorithms.jl:828-866) ------------------------------------
@pytest.mark.parametrize("M", [1, 4])
def test_downconvert_and_accumulate_reference_test_body(g, M):
    """test/algorithms.jl:1438-1514: GPS L1 PRN 1, 2500 samples, 1500 Hz, phases 0, taps (-1, 0, 1).  The reference
    asserts `accum[:, :, 2] == ones` (prompt products: signal = code x carrier, so wipe-off x replica gives 1) and
    `sum(accum, dims = 1) == [1476 2500 1476]`."""
    import torch
    N, L = 2500, 3
    system = g.GPSL1(use_gpu=True)
    signal, fs = g.gen_signal(system, 1, 1500.0, N, num_ants=g.NumAnts(M))
    corr = g.EarlyPromptLateCorrelator(g.NumAnts(M), g.NumAccumulators(L))
    shifts = g.get_correlator_sample_shifts(system, corr, fs, 0.5)
    dev = signal.re.device
    acc_re, acc_im = torch.zeros((L, M, N), device=dev), torch.zeros((L, M, N), device=dev)
    car_re, car_im = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    dw_re, dw_im = torch.zeros((M, N), device=dev), torch.zeros((M, N), device=dev)
    g.downconvert_and_accumulate_strided(acc_re, acc_im, car_re, car_im, dw_re, dw_im, signal.re, signal.im, system,
                                         g.get_code_frequency(system), 1500.0, fs, 0.0, 0.0, N, shifts, 1)
    acc = (acc_re + 1j * acc_im).cpu().numpy()
    assert np.allclose(acc[1], np.ones((M, N)), atol=2e-6)                       # prompt products == 1
    assert np.allclose(acc.sum(axis=2), np.array([1476, 2500, 1476])[:, None], rtol=RTOL)
    # the carrier replica and the downconverted signal (== the code, aligned with the prompt replica: :1434)
    n = np.arange(N)
    th = 2 * np.pi * n * 1500.0 / fs
    car = (car_re + 1j * car_im).cpu().numpy()
    assert np.abs(car - np.exp(1j * th)).max() < 2e-6
    rep = oracle.gen_code_replica(oracle.codes("GPSL1", 1), 0, 1.023e6, fs, 0.0, 0, N)
    dw = (dw_re + 1j * dw_im).cpu().numpy()
    assert np.abs(dw - rep[None, :]).max() < 2e-6


def test_kernel_algorithm_2_fills_the_materialised_buffers(g):
    """kernel_algorithm(..., KernelAlgorithm{2330}) with REAL accum / carrier / downconverted buffers
    (test/algorithms.jl:452-596): result in phi, per-sample products in accum."""
    import torch
    N, M, L = 2500, 4, 3
    system = g.GPSL1(use_gpu=True)
    signal, fs = g.gen_signal(system, 1, 1500.0, N, num_ants=g.NumAnts(M))
    corr = g.EarlyPromptLateCorrelator(g.NumAnts(M), g.NumAccumulators(L))
    shifts = g.get_correlator_sample_shifts(system, corr, fs, 0.5)
    dev = signal.re.device
    z = lambda *sh: torch.zeros(sh, device=dev)  # noqa: E731
    accum_re, accum_im, phi_re, phi_im = z(L, M, N), z(L, M, N), z(L, M, 5), z(L, M, 5)
    car_re, car_im, dw_re, dw_im = z(N), z(N), z(M, N), z(M, N)
    g.kernel_algorithm(None, None, None, None, system.codes, g.get_code_frequency(system), fs, 0.0, 1, N, 2,
                       g.get_code_length(system), accum_re, accum_im, phi_re, phi_im, car_re, car_im, dw_re, dw_im,
                       signal.re, signal.im, shifts, 1500.0, 0.0, g.NumAnts(M), None, g.KernelAlgorithm(2330))
    phi = (phi_re[:, :, 0] + 1j * phi_im[:, :, 0]).cpu().numpy()
    assert np.allclose(phi, np.array([1476, 2500, 1476])[:, None], rtol=RTOL)
    assert np.allclose((accum_re + 1j * accum_im).cpu().numpy().sum(axis=2), phi, rtol=RTOL)
    assert float(car_re.abs().max()) > 0.99 and float(dw_re.abs().max()) > 0.99


def test_downconvert_and_accumulate_matches_the_fused_kernel(g):
    """Random scenario: the column sums of the materialised products equal the fused correlator's output."""
    import torch
    case = make_case(31, system="GPSL5", N=9001, M=3, L=5, K=1, B=1, fs=25e6, if_hz=2.3e5)
    ref = oracle_result(case)[0, 0]
    ctx = g.get_context()
    sysobj = g.GPSL5(use_gpu=True)
    p = case["prm"][0, 0]
    re, im = torch.from_numpy(case["re"]).to(ctx.device), torch.from_numpy(case["im"]).to(ctx.device)
    acc_re = torch.zeros((5, 3, 9001), device=ctx.device)
    acc_im = torch.zeros_like(acc_re)
    g.downconvert_and_accumulate_strided(acc_re, acc_im, None, None, None, None, re, im, sysobj, p["code_freq_hz"],
                                         p["carrier_freq_hz"], case["fs"], p["code_phase_chips"],
                                         p["carrier_phase_cycles"], 9001, case["shifts"], int(p["prn0"]) + 1)
    got = (acc_re.double().sum(dim=2) + 1j * acc_im.double().sum(dim=2)).cpu().numpy()
    check_close(got[None, None], ref[None, None], what="materialised products, summed")


# ---- A3 (_nsat_): several satellites' replicas in one launch (src/algorithms.jl:78-98) ------------------------------
def test_gen_code_replica_nsat_bit_exact(g):
    import torch
    system = g.GPSL5(use_gpu=True)
    fs, n, first = 50e6, 50008, -4
    prns = np.array([1, 7, 12, 31])
    fc = 10.23e6 * (1 + np.array([0.0, 2e-6, -3e-6, 1e-6]))
    tau = np.array([0.0, 10229.75, 5000.5, 17.125])
    ctx = g.get_context()
    rep = torch.full((4, n + 8), -7.0, dtype=torch.float32, device=ctx.device)
    g.gen_code_replica_nsat(rep, system, fc, fs, tau, n, first, prns)
    got = rep.cpu().numpy()
    codes = oracle.codes("GPSL5", 32)
    for k in range(4):
        want = oracle.gen_code_replica(codes, int(prns[k]) - 1, fc[k], fs, tau[k], first, n)
        assert np.array_equal(got[k, :n], want), f"channel {k}"
        assert (got[k, n:] == -7.0).all()  # nothing written past `count`
    # and it equals the one-satellite operator called per channel
    one = torch.zeros(n, dtype=torch.float32, device=ctx.device)
    ctx.gen_code_replica(one, n, int(prns[2]) - 1, fc[2], fs, tau[2], first)
    assert torch.equal(one, rep[2, :n])


def test_noisy_generator_steering_noise_statistics_and_parity(g):
    """gat_gen_signal_noisy (SURVEY section 8-d build additions; the reference's generator is noise-free with identical
    antennas, src/gen_signal.jl:86-90): (1) sigma = 0 + steering == the noise-free signal rotated per antenna; (2) the noise
    is zero-mean white Gaussian of the requested sigma, a pure function of the seed (same seed -> same bits, in another
    layout too), different per antenna and per seed; (3) the correlator on the noisy multi-antenna signal still matches the
    FP64 oracle run on the very same samples."""
    import torch
    import oracle
    system = g.GPSL1(use_gpu=True)
    N, M, B, K = 20000, 4, 3, 2
    fs = N / 1e-3
    prm = g.make_params(np.array([[2, 9]] * B), 1.023e6, np.array([[1500.0, -2300.0]] * B), np.array([[100.25, 811.5]] * B), 0.0)
    steer = np.array([0.0, 0.125, 0.5, 0.71], dtype=np.float32)
    base_re, base_im = g.gen_signal_stream(system, prm, fs, N, M)
    st_re, st_im = g.gen_signal_stream(system, prm, fs, N, M, steering_cycles=steer)
    rot = np.exp(2j * np.pi * steer.astype(np.float64))[:, None]
    want = (base_re.cpu().numpy() + 1j * base_im.cpu().numpy()) * rot
    got = st_re.cpu().numpy() + 1j * st_im.cpu().numpy()
    assert np.abs(got - want).max() < 2e-6
    sigma = 1.5
    n1 = g.gen_signal_stream(system, prm, fs, N, M, steering_cycles=steer, noise_sigma=sigma, seed=7)
    n2 = g.gen_signal_stream(system, prm, fs, N, M, steering_cycles=steer, noise_sigma=sigma, seed=7)
    n3 = g.gen_signal_stream(system, prm, fs, N, M, steering_cycles=steer, noise_sigma=sigma, seed=8)
    assert torch.equal(n1[0], n2[0]) and torch.equal(n1[1], n2[1])
    assert not torch.equal(n1[0], n3[0])
    noise = (n1[0].cpu().numpy() + 1j * n1[1].cpu().numpy()) - got
    assert abs(noise.real.mean()) < 0.02 and abs(noise.imag.mean()) < 0.02
    assert abs(noise.real.std() - sigma) < 0.02 and abs(noise.imag.std() - sigma) < 0.02
    assert abs(np.corrcoef(noise.real.ravel(), noise.imag.ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(noise[0].real, noise[1].real)[0, 1]) < 0.02            # antennas: independent noise
    assert abs(np.corrcoef(noise[0].real[:-1], noise[0].real[1:])[0, 1]) < 0.02    # white
    kurt = np.mean(noise.real ** 4) / sigma ** 4
    assert abs(kurt - 3.0) < 0.1                                                    # Gaussian
    il = g.gen_signal_stream(system, prm, fs, N, M, layout=g.GAT_LAYOUT_INTERLEAVED, steering_cycles=steer, noise_sigma=sigma, seed=7)[0]
    assert torch.equal(il[..., 0], n1[0]) and torch.equal(il[..., 1], n1[1])        # same noise in another layout
    # parity on the noisy signal
    corr = g.EarlyPromptLateCorrelator(g.NumAnts(M), g.NumAccumulators(3))
    shifts = g.get_correlator_sample_shifts(system, corr, fs, 0.5)
    op = g.StreamCorrelator(system, N, M, B, K, shifts, fs)
    op.set_params(prm)
    op(n1[0], n1[1])
    oprm = oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"], prm["code_phase_chips"], prm["carrier_phase_cycles"])
    ref = oracle.correlate_f64(n1[0].cpu().numpy(), n1[1].cpu().numpy(), system.codes, oprm, fs, shifts, N=N)
    res = op.result()
    err = np.max(np.abs(res - ref) / np.abs(ref).max(axis=(2, 3), keepdims=True))
    assert err <= 1e-5, err
    # each antenna's prompt carries its steering phase
    ph = np.angle(res[0, 0, 1, :] / res[0, 0, 1, 0]) / (2 * np.pi)
    assert np.allclose((ph - steer + 0.5) % 1.0 - 0.5, 0.0, atol=0.01)
