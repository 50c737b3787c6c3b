"""GPU parity tests: the HIP path (through the C ABI) against the FP64 oracle on the same seeded
inputs, the reference's known answers, and size-independent properties.  Run with -m gpu."""
import zlib

import numpy as np
import pytest

import oracle
from tests.helpers import RTOL, check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gat():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def run_hip(g, case, layout=0, flags=0, start_misalign=0):
    import torch
    sysobj = g.GNSSDICT[case["system"]](use_gpu=True)
    dev = g.get_context().device
    N, M, B, K = case["N"], case["M"], case["B"], case["K"]
    re = torch.from_numpy(case["re"]).to(dev)
    im = torch.from_numpy(case["im"]).to(dev)
    if start_misalign:  # force the scalar-load kernel: shift the planes by a non-multiple of 4
        pad = torch.zeros((M, start_misalign), dtype=torch.float32, device=dev)
        re = torch.cat([pad, re], dim=1)[:, start_misalign:]
        im = torch.cat([pad, im], dim=1)[:, start_misalign:]
    op = g.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"], flags=flags)
    prm = g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                        case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"])
    op.set_params(prm)
    if layout == 0:
        op(re, im)
        op.last_signal = (re, im)  # kept for tests that launch again on the same buffers
    else:
        x = torch.stack([re, im], dim=-1).contiguous()
        op(x, None)
    return op.result(), op


# ---- the reference's own known answers (test/algorithms.jl:85, :191, :300, :1374, :1513) -------
@pytest.mark.parametrize("N,M,expect", [(2500, 1, [1476, 2500, 1476]), (2500, 4, [1476, 2500, 1476]),
                                        (2048, 4, [1024, 2048, 1024])])
def test_known_answer_operator_surface(gat, N, M, expect):
    g = gat
    system = g.GPSL1(use_gpu=True)
    signal, fs = g.gen_signal(system, 1, 1500.0, N, num_ants=g.NumAnts(M))
    correlator = g.EarlyPromptLateCorrelator(g.NumAnts(M), g.NumAccumulators(3))
    shifts = g.get_correlator_sample_shifts(system, correlator, fs, 0.5)
    assert list(shifts) == [-1, 0, 1]
    out = g.downconvert_and_correlate(system, signal, correlator, None, 0.0, None, 0.0, None,
                                      g.get_code_frequency(system), shifts, 1500.0, fs, 1, N, 1)
    acc = g.get_accumulators(out)  # [L, M]
    for m in range(M):
        # reference asserts isapprox with rtol sqrt(eps(Float32)); we hold 1e-5
        assert np.allclose(acc[:, m], np.array(expect, dtype=np.complex64), rtol=RTOL, atol=RTOL * N)


@pytest.mark.parametrize("name", ["1_3_cplx_multi", "1_4_cplx_multi_textmem", "2_3_cplx_multi",
                                  "3_4_cplx_multi_textmem", "4_4_cplx_multi_textmem"])
def test_kernel_algorithm_forms(gat, name):
    """The reference's test bodies (test/algorithms.jl:1-88, :308-447, :895-1027, :1029-1157) with
    their positional argument lists."""
    import torch
    g = gat
    N, M, L = 2500, 4, 3
    system = g.GPSL1(use_gpu=True)
    signal, fs = g.gen_signal(system, 1, 1500.0, N, num_ants=g.NumAnts(M))
    correlator = g.EarlyPromptLateCorrelator(g.NumAnts(M), g.NumAccumulators(L))
    shifts = g.get_correlator_sample_shifts(system, correlator, fs, 0.5)
    nshift = int(shifts[-1] - shifts[0])
    alg = g.KernelAlgorithm(g.ALGODICT[name])
    dev = signal.re.device
    blocks = 10
    buf = g.StructSignal(torch.zeros((L, M, blocks), device=dev), torch.zeros((L, M, blocks), device=dev))
    common = (None, None, None, None, system.codes, g.get_code_frequency(system), fs, 0.0, 1, N, nshift,
              g.get_code_length(system))
    tail = (None, None, None, None, signal.re, signal.im, shifts, 1500.0, 0.0, g.NumAnts(M), None, alg)
    if alg.id in (1330, 1331, 1431):
        g.kernel_algorithm(*common, buf, *tail)
    elif alg.id // 1000 == 2:
        big = torch.zeros((1,), device=dev)
        g.kernel_algorithm(*common, big, big, buf.re, buf.im, *tail)
    else:
        g.kernel_algorithm(*common, buf.re, buf.im, *tail)
    acc = (buf.re[:, :, 0] + 1j * buf.im[:, :, 0]).cpu().numpy()
    for m in range(M):
        assert np.allclose(acc[:, m], [1476, 2500, 1476], rtol=RTOL, atol=RTOL * N)


# ---- randomised parity against the FP64 oracle ---------------------------------------------------
GRID = [
    # system, N, M, L, K, B, if_hz
    ("GPSL1", 4000, 1, 3, 1, 1, 0.0),          # BASELINE config 1
    ("GPSL1", 20000, 4, 3, 1, 3, 0.0),         # BASELINE config 2 shape (3 blocks)
    ("GPSL1", 20000, 4, 3, 1, 2, 4.3e6),       # MHz-range IF
    ("GPSL5", 50000, 4, 5, 3, 1, 0.0),         # BASELINE config 3 shape (3 of 12 PRNs)
    ("GPSL1", 50000, 16, 3, 4, 1, 0.0),        # BASELINE config 4 per-GPU shape
    ("GPSL1", 2500, 3, 7, 2, 2, 1.0e5),        # odd antenna count, 7 taps
    ("GPSL1", 1023, 2, 9, 1, 2, 0.0),          # > 8 taps: two tap tiles
    ("GPSL1", 257, 5, 1, 1, 4, 0.0),           # N just over one wave row, single tap
    ("GPSL1", 7, 1, 3, 1, 1, 0.0),             # tiny
    ("GPSL5", 32768, 1, 3, 2, 1, 1.2e6),
]


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("cfg", GRID, ids=[f"{c[0]}-N{c[1]}-M{c[2]}-L{c[3]}-K{c[4]}-B{c[5]}" for c in GRID])
def test_parity_grid(gat, cfg, layout):
    system, N, M, L, K, B, if_hz = cfg
    case = make_case(zlib.crc32(repr(cfg).encode()), system=system, N=N, M=M, L=L, K=K, B=B, if_hz=if_hz)
    got, _ = run_hip(gat, case, layout=layout)
    check_close(got, oracle_result(case), what=str(cfg))


@pytest.mark.parametrize("cfg", [("GPSL1", 2048, 1, 3, 1, 1), ("GPSL1", 2048, 4, 7, 1, 1), ("GPSL1", 16384, 4, 3, 1, 1),
                                 ("GPSL1", 5000, 2, 3, 2, 2), ("GPSL1", 5000, 2, 3, 5, 1), ("GPSL5", 20000, 4, 5, 3, 1),
                                 ("GPSL1", 20000, 16, 3, 4, 1), ("GPSL1", 4099, 1, 9, 3, 1)],
                         ids=lambda c: f"{c[0]}-N{c[1]}-M{c[2]}-L{c[3]}-K{c[4]}-B{c[5]}")
def test_host_parameter_calls(gat, cfg):
    """gat_downconvert_and_correlate (HOST parameter records -- the reference's call shape, scalars per call): up to four
    records travel inside the kernel arguments (no upload in front of the launch), more are uploaded; a matrix-core kernel
    that takes an inline call uploads them after all.  Every route must give the device-parameter call's result bit for bit."""
    import torch
    from gpuacceleratedtracking_amd import _lib
    system, N, M, L, K, B = cfg
    case = make_case(zlib.crc32(repr(("host", cfg)).encode()), system=system, N=N, M=M, L=L, K=K, B=B)
    ctx = gat.get_context()
    dev = ctx.device
    ctx.set_codes(gat.GNSSDICT[system](use_gpu=True).codes)
    re = torch.from_numpy(case["re"]).to(dev).contiguous()
    im = torch.from_numpy(case["im"]).to(dev).contiguous()
    assert re.shape == (M, B * N)
    desc = _lib.SignalDesc(re.data_ptr(), im.data_ptr(), gat.GAT_LAYOUT_PLANAR, M, N, B * N, N, 0)
    p = case["prm"]
    prm = gat.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    prm = np.ascontiguousarray(prm.reshape(-1))
    prm_dev = torch.from_numpy(prm.view(np.uint8).copy()).to(dev)
    outs = []
    for params in (prm, prm_dev):
        o_re = torch.full((B * K * L * M,), float("nan"), dtype=torch.float32, device=dev)
        o_im = torch.full_like(o_re, float("nan"))
        ctx.downconvert_and_correlate(desc, params, B, K, case["shifts"], case["fs"], o_re, o_im)
        torch.cuda.synchronize()
        outs.append((o_re.cpu().numpy(), o_im.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    got = (outs[0][0] + 1j * outs[0][1]).reshape(B, K, L, M)
    ref = oracle_result(case)
    check_close(got, ref, what=str(cfg))
    # the same host records through the atomic second stage (memsets + launch; summation order is not fixed: tolerance only)
    o_re = torch.full((B * K * L * M,), float("nan"), dtype=torch.float32, device=dev)
    o_im = torch.full_like(o_re, float("nan"))
    ctx.downconvert_and_correlate(desc, prm, B, K, case["shifts"], case["fs"], o_re, o_im, flags=gat.GAT_FLAG_ATOMIC)
    torch.cuda.synchronize()
    check_close((o_re.cpu().numpy() + 1j * o_im.cpu().numpy()).reshape(B, K, L, M), ref, what=f"atomic {cfg}")


@pytest.mark.parametrize("N", [1, 3, 255, 256, 1021, 1025, 4099])
def test_ragged_lengths_and_scalar_path(gat, N):
    """N not a multiple of the vector width / workgroup chunk, on both the 16-byte-vector kernel
    and the scalar-load kernel (unaligned plane base)."""
    case = make_case(1000 + N, N=N, M=2, L=3, K=1, B=3, fs=4.0e6)
    ref = oracle_result(case)
    for mis in (0, 1, 3):
        got, op = run_hip(gat, case, start_misalign=mis)
        # 16-byte vectors need every block start aligned: plane base and block_stride (= N) % 4
        assert op.ctx.last_launch_info()["vec"] == (4 if (mis == 0 and N % 4 == 0) else 1)
        check_close(got, ref, what=f"N={N} misalign={mis}")


def test_vector_and_scalar_kernels_selected(gat):
    case = make_case(5, N=4096, M=4, L=3, B=2)
    _, op = run_hip(gat, case)
    assert op.ctx.last_launch_info()["vec"] == 4
    _, op = run_hip(gat, case, start_misalign=1)
    assert op.ctx.last_launch_info()["vec"] == 1


def test_split_and_finalize_path_deterministic(gat):
    """B = 1 splits one block over many workgroups (two-stage sum): must equal the oracle and be
    bit-identical across repeated launches."""
    case = make_case(77, N=200000, M=4, L=3, K=1, B=1, fs=200e6)
    got1, op = run_hip(gat, case)
    info = op.ctx.last_launch_info()
    assert info["splits"] > 1 and info["finalize_launched"] == 1
    check_close(got1, oracle_result(case), what="split")
    got2, _ = run_hip(gat, case)
    assert np.array_equal(got1.view(np.float32), got2.view(np.float32))


@pytest.mark.parametrize("N,M,K,B,L", [(120000, 16, 4, 3, 3), (65536 + 4, 4, 3, 5, 5), (300000, 1, 1, 1, 3), (49996, 2, 7, 2, 7),
                                       (30000, 16, 9, 2, 3)])
def test_split_second_stage_soak(gat, N, M, K, B, L):
    """Split blocks (two-stage sum, one thread per output element in the second stage): many groups, ragged last splits,
    channel-looping and antenna-parallel tilings -- equal to the oracle, and forty repeated launches on the same buffers
    bit-identical."""
    import torch
    g = gat
    case = make_case(900 + M + K, N=N, M=M, K=K, B=B, L=L, fs=N / 1e-3)
    got, op = run_hip(g, case)
    info = op.ctx.last_launch_info()
    assert info["matrix_core"] == 0 and info["splits"] > 1 and info["finalize_launched"] == 1, info
    check_close(got, oracle_result(case), what="split second stage")
    ref_re, ref_im = op.out_re.clone(), op.out_im.clone()
    for i in range(40):
        op.out_re.fill_(float("nan"))
        op.out_im.fill_(float("nan"))
        op(*op.last_signal)
        assert torch.equal(op.out_re, ref_re) and torch.equal(op.out_im, ref_im), i


def test_atomic_mode(gat):
    import gpuacceleratedtracking_amd as g
    case = make_case(78, N=100000, M=4, L=3, K=2, B=1, fs=100e6)
    got, op = run_hip(gat, case, flags=g.GAT_FLAG_ATOMIC)
    check_close(got, oracle_result(case), what="atomic")
    # outputs are overwritten, not accumulated across calls (reference defect D4 not reproduced)
    got2, _ = run_hip(gat, case, flags=g.GAT_FLAG_ATOMIC)
    check_close(got2, oracle_result(case), what="atomic second call")


def test_absent_prn_and_noise(gat):
    """Channel whose PRN is NOT in the signal (small |R|) plus AWGN: error is bounded relative to
    full scale N (a relative bound on a near-zero accumulator is meaningless in FP32)."""
    case = make_case(79, N=20000, M=4, L=3, K=1, B=2, noise=1.0)
    case["prm"]["prn0"] = (case["prm"]["prn0"] + 5) % 32
    got, _ = run_hip(gat, case)
    ref = oracle_result(case)
    assert np.abs(got - ref).max() <= RTOL * case["N"]


def test_l5_chip_edges_exact(gat):
    """GPS L5 at 50 MHz with tau = 0: fc/fs*n hits integers at n = 5000k, so a fused
    multiply-add (or float phase) would move chip edges; all-ones signal makes R an exact integer
    sum of chips -> must match the oracle EXACTLY."""
    import torch
    g = gat
    N, M = 50000, 1
    system = g.GPSL5(use_gpu=True)
    dev = g.get_context().device
    re = torch.ones((M, N), device=dev)
    im = torch.zeros((M, N), device=dev)
    shifts = np.array([-4, -2, 0, 2, 4], dtype=np.int32)
    op = g.StreamCorrelator(system, N, M, 1, 1, shifts, 50e6)
    op.set_params(g.make_params(0, 10.23e6, 0.0, 0.0, 0.0, shape=(1, 1)))
    op(re, im)
    got = op.result()
    prm = oracle.make_params(0, 10.23e6, 0.0, 0.0, 0.0, shape=(1, 1))
    ref = oracle.correlate_f64(np.ones((M, N), np.float32), np.zeros((M, N), np.float32), system.codes, prm,
                               50e6, shifts)
    assert np.array_equal(got.real.astype(np.int64), np.rint(ref.real).astype(np.int64))
    assert np.abs(got.imag).max() == 0.0


# ---- stand-alone operators ---------------------------------------------------------------------
@pytest.mark.parametrize("system,fs,tau", [("GPSL1", 2.5e6, 0.0), ("GPSL1", 20e6, 511.75), ("GPSL5", 50e6, 10229.5)])
def test_gen_code_replica_bit_exact(gat, system, fs, tau):
    import torch
    g = gat
    sysobj = g.GNSSDICT[system](use_gpu=True)
    fc = g.get_code_frequency(sysobj)
    N = 20000
    corr = g.EarlyPromptLateCorrelator(1, 3)
    shifts = g.get_correlator_sample_shifts(sysobj, corr, fs, 0.5)
    count = N + int(shifts[-1] - shifts[0])
    rep = torch.zeros(count + 8, device=g.get_context().device)
    g.gen_code_replica(rep, sysobj, fc, fs, tau, 1, N, shifts, 3)
    ref = oracle.gen_code_replica(sysobj.codes, 2, fc, fs, tau, int(shifts[0]), count)
    got = rep.cpu().numpy()
    assert np.array_equal(got[:count], ref)
    assert (got[count:] == 0).all()


def test_gen_signal_matches_oracle(gat):
    g = gat
    system = g.GPSL1(use_gpu=True)
    N, M = 20000, 4
    signal, fs = g.gen_signal(system, 7, 1500.0, N, num_ants=g.NumAnts(M), start_code_phase=100.25,
                              start_carrier_phase=0.3)
    re, im = signal.cpu()
    ore, oim = oracle.gen_signal(system.codes, 6, 1.023e6, fs, 1500.0, 100.25, 0.3, N, M)
    # same Float32-rounded phase, cos/sin differ by at most a few ulp between libm and OCML
    assert np.abs(re - ore).max() <= 1e-6 and np.abs(im - oim).max() <= 1e-6
    assert np.array_equal(re[0], re[3])  # identical antennas (src/gen_signal.jl:89-90)


@pytest.mark.parametrize("n,shape", [(2048, (3, 1)), (2500, (3, 4)), (32768, (1, 16)), (100000, (3, 4))])
def test_reduce_cplx_multi_all_ones(gat, n, shape):
    """test/reduction.jl:51-52 and siblings: ones + 0im of size (N, M, L) reduces to [N N N]."""
    import torch
    g = gat
    dev = g.get_context().device
    re = torch.ones(shape + (n,), device=dev)
    im = torch.zeros(shape + (n,), device=dev)
    o_re, o_im = g.reduce_cplx_multi(re, im)
    assert (o_re.cpu().numpy() == n).all() and (o_im.cpu().numpy() == 0).all()


def test_reduce_cplx_multi_random(gat):
    import torch
    g = gat
    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 7, 33333)).astype(np.float32)
    b = rng.standard_normal((5, 7, 33333)).astype(np.float32)
    dev = g.get_context().device
    o_re, o_im = g.reduce_cplx_multi(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev))
    ref = oracle.reduce_cplx_multi(a.reshape(35, -1), b.reshape(35, -1)).reshape(5, 7)
    scale = np.abs(a).sum(axis=-1).max()
    assert np.abs(o_re.cpu().numpy() - ref.real).max() <= 1e-6 * scale
    assert np.abs(o_im.cpu().numpy() - ref.imag).max() <= 1e-6 * scale


def test_per_channel_signal_3d(gat):
    """downconvert_and_correlate_kernel_3d_4431! semantics (src/algorithms.jl:637-718): one signal
    per satellite, signal[n, m, k]."""
    import torch
    g = gat
    N, M, K = 2048, 4, 3
    system = g.GPSL1(use_gpu=True)
    signal, fs = g.gen_signal(system, [1, 2, 3], 1500.0, N, num_ants=g.NumAnts(M))
    assert tuple(signal.re.shape) == (K, M, N)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    op = g.StreamCorrelator(system, N, M, 1, K, shifts, fs, per_channel_signal=True)
    op.set_params(g.make_params(np.arange(K), 1.023e6, 1500.0, 0.0, 0.0, shape=(1, K)))
    op(signal.re, signal.im)
    got = op.result()
    for k in range(K):
        for m in range(M):
            assert np.allclose(got[0, k, :, m], [1024, 2048, 1024], rtol=RTOL, atol=RTOL * N)


# ---- error behaviour: fail loudly, never fall back ------------------------------------------------
def test_errors(gat):
    import torch
    g = gat
    system = g.GPSL1(use_gpu=True)
    N, M = 1000, 2
    signal, fs = g.gen_signal(system, 1, 0.0, N, num_ants=M)
    corr = g.EarlyPromptLateCorrelator(M, 3)
    shifts = g.get_correlator_sample_shifts(system, corr, fs)
    with pytest.raises(ValueError):
        g.downconvert_and_correlate(system, signal, corr, None, 0.0, None, 0.0, None, 1.023e6, shifts, 0.0, fs, 1, N, 33)
    with pytest.raises(ValueError):
        g.downconvert_and_correlate(system, signal, corr, None, 0.0, None, 0.0, None, 1.023e6, shifts, 0.0, fs, 2, N, 1)
    op = g.StreamCorrelator(system, N, M, 1, 1, shifts, fs)
    with pytest.raises(RuntimeError):
        op(signal.re, signal.im)  # params not set
    op.set_params(g.make_params(0, 1.023e6, 0.0, 0.0, 0.0, shape=(1, 1)))
    with pytest.raises(g.GatError):
        bad = g.StreamCorrelator(system, N, M, 1, 1, np.arange(40), fs)  # > GAT_MAX_TAPS
        bad.set_params(g.make_params(0, 1.023e6, 0.0, 0.0, 0.0, shape=(1, 1)))
        bad(signal.re, signal.im)
    with pytest.raises(g.GatError):
        op.ctx.downconvert_and_correlate(op.describe(signal.re, signal.im), op.params_dev, 1, 1, shifts, -1.0,
                                         op.out_re, op.out_im)
    # the host entry point mirrors the kernels' `bad` predicate: a negative code frequency is GAT_ERR_RANGE up
    # front, not GAT_OK plus NaN outputs
    host_prm = g.make_params(0, -1.023e6, 0.0, 0.0, 0.0, shape=(1, 1))
    with pytest.raises(g.GatError) as ei:
        op.ctx.downconvert_and_correlate(op.describe(signal.re, signal.im), host_prm, 1, 1, shifts, fs, op.out_re, op.out_im)
    assert ei.value.status == 2
    # ... while device-resident parameters cannot be checked on the host: the kernel poisons that channel with NaN
    op.params_dev = op.ctx.params_to_device(host_prm)
    op._prepared = None
    op(signal.re, signal.im)
    assert np.isnan(op.result().view(np.float32)).all()
    # before gat_set_codes: GAT_ERR_STATE, whatever the prn says
    fresh = g.Context(0, torch.cuda.Stream())
    with pytest.raises(g.GatError) as ei:
        fresh.downconvert_and_correlate(op.describe(signal.re, signal.im), g.make_params(99, 1.023e6, 0.0, 0.0, 0.0, shape=(1, 1)),
                                        1, 1, shifts, fs, op.out_re, op.out_im)
    assert ei.value.status == 3
    fresh.close()
    with pytest.raises(NotImplementedError):
        g.run_kernel_benchmark({"processor": "CPU", "GNSS": "GPSL1", "num_samples": 2048, "num_ants": 1,
                                "num_correlators": 3, "algorithm": "1_3_cplx_multi"})


def test_run_kernel_benchmark(gat):
    g = gat
    r = g.run_kernel_benchmark({"processor": "GPU", "GNSS": "GPSL1", "num_samples": 2048, "num_ants": 4,
                                "num_correlators": 3, "algorithm": "4_4_cplx_multi_textmem"}, seconds=0.2)
    for key in ("RawTimes", "Minimum", "Median", "Mean", "σ", "Maximum", "os", "CPU_model", "GPU_model", "HIP",
                "algorithm"):
        assert key in r
    assert r["algorithm"] == "4_4_cplx_multi_textmem" and r["Minimum"] > 0
    assert np.allclose(r["accumulators"][:, 0], [1024, 2048, 1024], rtol=1e-5)


# ---- "next" row 1: packed integer IF samples (int16 / int8 {re, im} pairs) -------------------------
def _quantised_case(seed, dtype, amp, **kw):
    case = make_case(seed, **kw)
    lim = np.iinfo(dtype)
    q_re = np.clip(np.rint(case["re"] * amp), lim.min, lim.max).astype(dtype)
    q_im = np.clip(np.rint(case["im"] * amp), lim.min, lim.max).astype(dtype)
    case["re"], case["im"] = q_re.astype(np.float32), q_im.astype(np.float32)  # what the oracle sees (exact)
    return case, np.stack([q_re, q_im], axis=-1)  # [M, B*N, 2]


@pytest.mark.parametrize("dtype,amp", [(np.int16, 3000.0), (np.int8, 25.0)])
@pytest.mark.parametrize("cfg", [("GPSL1", 20000, 4, 3, 1, 3), ("GPSL1", 4099, 2, 3, 2, 2), ("GPSL5", 32768, 1, 5, 2, 1),
                                 ("GPSL1", 8, 3, 7, 1, 2)], ids=lambda c: f"{c[0]}-N{c[1]}-M{c[2]}-L{c[3]}-K{c[4]}-B{c[5]}")
def test_int_ingest_parity(gat, cfg, dtype, amp):
    import torch
    g = gat
    system, N, M, L, K, B = cfg
    case, q = _quantised_case(zlib.crc32(repr(cfg).encode()), dtype, amp / K, system=system, N=N, M=M, L=L, K=K, B=B)
    ref = oracle_result(case)
    dev = g.get_context().device
    sysobj = g.GNSSDICT[system](use_gpu=True)
    prm = g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                        case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"])
    for mis in (0, 1):
        x = torch.from_numpy(q).to(dev)
        if mis:
            pad = torch.zeros((M, mis, 2), dtype=x.dtype, device=dev)
            x = torch.cat([pad, x], dim=1)[:, mis:]
        op = g.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"])
        op.set_params(prm)
        op(x, None)
        spv = 4 if dtype == np.int16 else 8
        assert op.ctx.last_launch_info()["vec"] == (4 if (mis == 0 and N % spv == 0) else 1)
        check_close(op.result(), ref, what=f"{cfg} {dtype.__name__} mis={mis}")


def test_gen_signal_int_layouts(gat):
    """gat_gen_signal in the integer layouts == rint(amplitude * float signal), saturated."""
    g = gat
    system = g.GPSL1(use_gpu=True)
    N, M, B = 5000, 2, 2
    prm = g.make_params([[3], [7]], 1.023e6, [[1500.0], [-900.0]], [[10.5], [700.25]], [[0.3], [1.1]], shape=(B, 1))
    f_re, f_im = g.gen_signal_stream(system, prm, 5e6, N, M)
    for layout, amp, lim in ((g.GAT_LAYOUT_INTERLEAVED_I16, 2000.0, 32767), (g.GAT_LAYOUT_INTERLEAVED_I8, 300.0, 127)):
        x, _ = g.gen_signal_stream(system, prm, 5e6, N, M, layout=layout, amplitude=amp)
        want_re = np.clip(np.rint(f_re.cpu().numpy() * np.float32(amp)), -lim - 1, lim)
        want_im = np.clip(np.rint(f_im.cpu().numpy() * np.float32(amp)), -lim - 1, lim)
        got = x.cpu().numpy()
        assert np.array_equal(got[..., 0], want_re) and np.array_equal(got[..., 1], want_im)
    xf, _ = g.gen_signal_stream(system, prm, 5e6, N, M, layout=g.GAT_LAYOUT_INTERLEAVED)
    assert np.array_equal(xf.cpu().numpy()[..., 0], f_re.cpu().numpy())


# ---- taps in any order / wider than the LDS replica segment: host-side tap grouping ----------------
@pytest.mark.parametrize("shifts", [[5, -3000, 0, 2500, -1, 7], [3, 2, 1, 0, -1, -2, -3, 4, 5, 6, -7], [0], [40000, -40000],
                                    [-900, -301, 0, 300, 901],   # span 1801: one launch with a replica segment sized for it
                                    [1024, -1024, 0]])            # span 2048: the widest single launch
def test_unsorted_and_wide_taps(gat, shifts):
    case = make_case(31 + len(shifts), N=12000, M=2, L=3, K=2, B=2, fs=8e6)
    case["shifts"] = np.asarray(shifts, dtype=np.int32)
    case["L"] = len(shifts)
    for layout in (0, 1):
        got, _ = run_hip(gat, case, layout=layout)
        check_close(got, oracle_result(case), what=f"shifts={shifts}")


# ---- "next" row 3: code-phase error of the Float32 normalised-coordinate (texture) replica -----------
@pytest.mark.parametrize("N", [2048, 20000, 262144])
def test_f32_coordinate_replica_matches_its_definition(gat, N):
    """gat_gen_code_replica_f32coord against a numpy statement of the same Float32 arithmetic
    (src/algorithms.jl:121-140: phase / code_length as a normalised texture coordinate)."""
    import torch
    g = gat
    system = g.GPSL1(use_gpu=True)
    fs, fc, lc = N / 1e-3, 1.023e6, 1023
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(1, 3), fs, 0.5)
    count = N + int(shifts[-1] - shifts[0])
    rep = torch.zeros(count, device=g.get_context().device)
    g.gen_code_replica(rep, system, fc, fs, 0.0, 1, N, shifts, 1, texture_coordinates=True)
    i = np.arange(count, dtype=np.float64) + float(shifts[0])
    u = ((np.float64(fc) / np.float64(fs)) * i + 0.0) / np.float64(lc)
    u = u.astype(np.float32)
    w = u - np.floor(u)
    idx = np.clip(np.floor(w * np.float32(lc)).astype(np.int64), 0, lc - 1)
    assert np.array_equal(rep.cpu().numpy(), system.codes[0][idx].astype(np.float32))
    exact = oracle.gen_code_replica(system.codes, 0, fc, fs, 0.0, int(shifts[0]), count)
    err = np.abs(rep.cpu().numpy() - exact).sum() / N
    assert err < 0.05  # a small fraction of samples sits within float32 resolution of a chip edge


def test_c_abi_from_plain_c(gat):
    """examples/gat_known_answer.c: the C ABI driven from plain C (no Python objects, library-owned
    stream and memory) reproduces the reference's known answer."""
    import subprocess
    from gpuacceleratedtracking_amd import build
    exe = build.build_c_example()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK (known answer 1476 2500 1476)" in r.stdout
    assert "two-satellite replica: OK" in r.stdout  # the stage operators julia/GATHip.jl binds, same sequence in C
