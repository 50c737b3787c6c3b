"""Every workgroup tiling of the fused vector kernel (csrc/gat_dc.h) against the FP64 oracle.

The kernel family is templated on (antennas per wave, antenna-tile waves per workgroup AW, channels per workgroup
KT -- several only together with AW = 4); the host picks a tiling per shape and `gat_set_vector_tiling` caps it.  These tests force each cap
combination on the same seeded inputs -- ragged block ends, several short blocks per workgroup, channel counts that
do not divide KT (trailing invalid channel slots), every sample format -- and compare with the oracle at the north
star's 1e-5; all tilings must also agree with each other on chip edges (bit-exact replica => identical all-ones sums).
Run with -m gpu."""
import zlib

import numpy as np
import pytest

import oracle
from tests.helpers import check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gat():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


@pytest.fixture()
def vector_ctx(gat):
    ctx = gat.get_context()
    ctx.set_matrix_core(gat.GAT_MC_VECTOR)
    yield ctx
    ctx.set_matrix_core(gat.GAT_MC_AUTO)
    ctx.set_vector_tiling(4, 4, 16)


def run_case(g, ctx, case, layout=0, scale=None):
    import torch
    sysobj = g.GNSSDICT[case["system"]](use_gpu=True)
    dev = ctx.device
    N, M, B, K = case["N"], case["M"], case["B"], case["K"]
    op = g.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"], ctx=ctx)
    op.set_params(g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                                case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"]))
    re, im = torch.from_numpy(case["re"]).to(dev), torch.from_numpy(case["im"]).to(dev)
    if layout == 0:
        op(re, im)
    elif layout == 1:
        op(torch.stack([re, im], dim=-1).contiguous(), None)
    else:
        dt = torch.int16 if layout == 2 else torch.int8
        op(torch.stack([re, im], dim=-1).to(dt).contiguous(), None)
    return op.result(), ctx.last_launch_info()


SHAPES = [
    # system, N, M, L, K, B
    ("GPSL1", 20000, 4, 3, 1, 3),    # configs[1] shape
    ("GPSL1", 5000, 16, 3, 4, 2),    # configs[3] shard shape (shortened): AW = 4, KT = 4
    ("GPSL1", 5003, 16, 3, 3, 2),    # ragged end, scalar tail, K = 3 -> one invalid channel slot at KT = 4
    ("GPSL5", 20000, 4, 5, 6, 1),    # configs[2] shape (6 PRNs)
    ("GPSL1", 4000, 1, 3, 5, 9),     # single antenna, 5 channels, 9 short blocks -> several blocks per workgroup
    ("GPSL1", 3000, 8, 3, 2, 3),     # two antenna tiles, one workgroup each
    ("GPSL1", 2047, 12, 7, 2, 2),    # 3 antenna tiles (AW = 1), 7 taps
    ("GPSL1", 700, 32, 1, 4, 2),     # 8 antenna tiles -> two antenna groups of AW = 4
    ("GPSL1", 333, 2, 8, 7, 4),      # two antennas per wave, 8 taps, 7 channels
]
TILINGS = [(1, 1, 1), (4, 1, 1), (1, 4, 1), (4, 4, 16), (2, 2, 4), (4, 2, 1)]


@pytest.mark.parametrize("tiling", TILINGS, ids=[f"aw{t[0]}-kt{t[1]}-bpw{t[2]}" for t in TILINGS])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"{s[0]}-N{s[1]}-M{s[2]}-L{s[3]}-K{s[4]}-B{s[5]}" for s in SHAPES])
def test_every_tiling_matches_the_oracle(gat, vector_ctx, shape, tiling):
    system, N, M, L, K, B = shape
    case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=B)
    vector_ctx.set_vector_tiling(*tiling)
    got, info = run_case(gat, vector_ctx, case)
    assert info["matrix_core"] == 0, info
    assert info["channels_per_wg"] <= tiling[1] and info["blocks_per_wg"] <= tiling[2], info
    check_close(got, oracle_result(case), what=f"{shape} {tiling} {info}")


@pytest.mark.parametrize("layout", [1, 2, 3])
@pytest.mark.parametrize("tiling", [(1, 1, 1), (4, 4, 16)])
def test_tilings_in_every_sample_format(gat, vector_ctx, layout, tiling):
    """Interleaved ComplexF32 / int16 / int8 input: integer-valued samples so that every format holds them exactly."""
    shape = ("GPSL1", 6000, 16, 3, 4, 2)
    system, N, M, L, K, B = shape
    case = make_case(77, system=system, N=N, M=M, L=L, K=K, B=B)
    amp = 100.0 if layout == 3 else 1000.0
    peak = max(np.abs(case["re"]).max(), np.abs(case["im"]).max())
    case["re"] = np.rint(case["re"] * (amp / peak)).astype(np.float32)
    case["im"] = np.rint(case["im"] * (amp / peak)).astype(np.float32)
    vector_ctx.set_vector_tiling(*tiling)
    got, info = run_case(gat, vector_ctx, case, layout=layout)
    assert info["matrix_core"] == 0
    check_close(got, oracle_result(case), what=f"layout {layout} {tiling}")


def test_chip_edges_identical_in_every_tiling_and_kernel(gat):
    """All-ones signal, zero carrier: every accumulator is an integer (sum of +-1 chips), so ANY chip-edge
    disagreement between the exact walk, the matrix kernels and the oracle shows up as a different integer.
    GPS L5 at 50 MHz (4.9 samples per chip, 5 taps) and a fractional code phase near an edge."""
    import torch
    g = gat
    ctx = g.get_context()
    N, M, K, L, B = 50000, 16, 4, 5, 2
    system = g.GPSL5(use_gpu=True)
    fs, fc = N / 1e-3, 10.23e6
    shifts = oracle.sample_shifts(L, fs, fc)
    rng = np.random.default_rng(5)
    tau = np.array([[0.0, 10229.999999, 5115.5, 1e-9], [3.0 - 1e-12, 7.25, 9000.125, 10229.5]])
    prm = oracle.make_params(np.broadcast_to(np.arange(K), (B, K)), fc * (1 + rng.uniform(-3e-6, 3e-6, (B, K))),
                             np.zeros((B, K)), tau, np.zeros((B, K)))
    re = np.ones((M, B * N), dtype=np.float32)
    im = np.zeros_like(re)
    ref = oracle.correlate_f64(re, im, oracle.codes("GPSL5", 32), prm, fs, shifts, N=N)
    assert np.all(ref.imag == 0) and np.all(ref.real == np.rint(ref.real))
    op = g.StreamCorrelator(system, N, M, B, K, shifts, fs, ctx=ctx)
    op.set_params(g.make_params(prm["prn0"], prm["code_freq_hz"], prm["carrier_freq_hz"], prm["code_phase_chips"],
                                prm["carrier_phase_cycles"]))
    d_re, d_im = torch.from_numpy(re).to(ctx.device), torch.from_numpy(im).to(ctx.device)
    try:
        for mode, tiling in [(g.GAT_MC_VECTOR, (1, 1, 1)), (g.GAT_MC_VECTOR, (4, 4, 16)), (g.GAT_MC_VECTOR, (4, 2, 1)),
                             (g.GAT_MC_F32, (4, 4, 16)), (g.GAT_MC_BF16_SPLIT, (4, 4, 16))]:
            ctx.set_matrix_core(mode)
            ctx.set_vector_tiling(*tiling)
            op(d_re, d_im)
            got = op.result()
            assert np.array_equal(got.real.astype(np.float64), ref.real), (mode, tiling, ctx.last_launch_info())
            assert np.all(got.imag == 0)
    finally:
        ctx.set_matrix_core(g.GAT_MC_AUTO)
        ctx.set_vector_tiling(4, 4, 16)


def test_exact_walk_survives_adversarial_code_phases(gat, vector_ctx):
    """Code phases chosen so that ratio*(n+shift)+tau lands within a few ulps of an integer at many samples (ratio =
    exactly 1/16 chip per sample, tau at and around multiples of 1/16): the walk's margin test must hand these to the
    exact evaluation.  All-ones signal => integer sums, compared exactly."""
    import torch
    g = gat
    N, M, K, L, B = 8192, 4, 4, 3, 3
    system = g.GPSL1(use_gpu=True)
    fc = 1.023e6
    fs = fc * 16.0          # ratio = 1/16 exactly
    shifts = np.array([-8, 0, 8], dtype=np.int32)
    eps = np.spacing(512.0)
    tau = np.array([[0.0, 0.0625, 511.9375, 1022.9375 + 0.0],
                    [512.0 - eps, 512.0 + eps, 0.0625 - np.spacing(0.0625), 0.0625 + np.spacing(0.0625)],
                    [1022.0 + 15 * 0.0625, 3 * 0.0625, 700.5, 1e-300]])
    prm = oracle.make_params(np.broadcast_to(np.arange(K), (B, K)), np.full((B, K), fc), np.zeros((B, K)), tau,
                             np.zeros((B, K)))
    re = np.ones((M, B * N), dtype=np.float32)
    im = np.zeros_like(re)
    ref = oracle.correlate_f64(re, im, oracle.codes("GPSL1", 32), prm, fs, shifts, N=N)
    op = g.StreamCorrelator(system, N, M, B, K, shifts, fs, ctx=vector_ctx)
    op.set_params(g.make_params(prm["prn0"], prm["code_freq_hz"], prm["carrier_freq_hz"], prm["code_phase_chips"],
                                prm["carrier_phase_cycles"]))
    d_re, d_im = torch.from_numpy(re).to(vector_ctx.device), torch.from_numpy(im).to(vector_ctx.device)
    for tiling in [(1, 1, 1), (4, 4, 16)]:
        vector_ctx.set_vector_tiling(*tiling)
        op(d_re, d_im)
        got = op.result()
        assert np.array_equal(got.real.astype(np.float64), ref.real), tiling


# ---- the two-channel 2 x 2 tile (round 5: two waves of two antennas, two channels each; option dc_aw2) ----------------------
TWO_BY_TWO_SHAPES = [
    # system, N, M, L, K, B
    ("GPSL1", 20000, 4, 3, 2, 3),    # configs[1]'s tile, two channels
    ("GPSL5", 20000, 4, 5, 6, 2),    # configs[2] family: long codes -> sign-bit tables, quads
    ("GPSL5", 6020, 4, 5, 3, 5),     # odd channel count (one empty channel slot), blocks that start off a 128-byte line
    ("GPSL1", 5004, 8, 3, 4, 2),     # two antenna groups, blocks that start 16 bytes into a 128-byte line
    ("GPSL1", 2048, 12, 7, 2, 2),    # three antenna groups, seven taps (three waves per SIMD)
    ("GPSL1", 332, 4, 8, 7, 4),      # eight taps, seven channels, blocks of less than one step
    ("GPSL1", 2500, 4, 3, 2, 1),     # the reference's N (2500 = 2.5 MHz: tap spacing 1 -> taps at ODD distances: the shifted copy, no quads)
    ("GPSL5", 40000, 4, 3, 12, 1),   # one block split over workgroups (second stage), twelve channels
]
FILL_MODES = [dict(), dict(dc_quads=0), dict(dc_bits=0), dict(dc_bits=2), dict(dc_bits=2, dc_quads=0), dict(dc_seg=2)]


@pytest.fixture()
def two_by_two_ctx(gat, vector_ctx):
    yield vector_ctx
    for name, val in (("dc_aw2", -1), ("dc_quads", -1), ("dc_bits", 1), ("dc_seg", 0)):
        vector_ctx.set_option(name, val)


@pytest.mark.parametrize("mode", FILL_MODES, ids=lambda m: "-".join(f"{k}{v}" for k, v in m.items()) or "default")
@pytest.mark.parametrize("shape", TWO_BY_TWO_SHAPES, ids=[f"{s[0]}-N{s[1]}-M{s[2]}-L{s[3]}-K{s[4]}-B{s[5]}" for s in TWO_BY_TWO_SHAPES])
def test_two_channel_two_by_two_tile_matches_the_oracle(gat, two_by_two_ctx, shape, mode):
    system, N, M, L, K, B = shape
    case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=B)
    ctx = two_by_two_ctx
    ctx.set_option("dc_aw2", 1)
    for name, val in mode.items():
        ctx.set_option(name, val)
    got, info = run_case(gat, ctx, case)
    assert info["matrix_core"] == 0 and info["channels_per_wg"] == 2 and info["ant_tile"] == 4 and info["threads"] == 256, info
    check_close(got, oracle_result(case), what=f"{shape} {mode} {info}")
    # the same call on the one-wave-of-four tile: the two tilings agree to rounding
    ctx.set_option("dc_aw2", 0)
    ref_tile, info0 = run_case(gat, ctx, case)
    assert info0["channels_per_wg"] == 1
    assert np.abs(got - ref_tile).max() <= 2e-6 * np.abs(ref_tile).max()


@pytest.mark.parametrize("layout", [1, 2])
def test_two_by_two_tile_in_complex_and_int16_samples(gat, two_by_two_ctx, layout):
    shape = ("GPSL5", 6000, 4, 5, 4, 3)
    system, N, M, L, K, B = shape
    case = make_case(78, system=system, N=N, M=M, L=L, K=K, B=B)
    peak = max(np.abs(case["re"]).max(), np.abs(case["im"]).max())
    case["re"] = np.rint(case["re"] * (1000.0 / peak)).astype(np.float32)
    case["im"] = np.rint(case["im"] * (1000.0 / peak)).astype(np.float32)
    two_by_two_ctx.set_option("dc_aw2", 1)
    got, info = run_case(gat, two_by_two_ctx, case, layout=layout)
    assert info["channels_per_wg"] == 2 and info["ant_tile"] == 4, info
    check_close(got, oracle_result(case), what=f"layout {layout} {info}")


def test_two_by_two_tile_is_the_planners_choice_where_it_measured_faster(gat, vector_ctx):
    """The rule of option dc_aw2 = -1 (gat_planner.cpp): float / int16 samples, four-antenna tiles that are no 16-antenna
    group, two or more channels on one signal, enough workgroups to fill the chip; never int8 pairs, never the latency regime."""
    import torch
    g = gat
    sysobj = g.GPSL1(use_gpu=True)
    shifts = np.array([-10, 0, 10], dtype=np.int32)

    def tile(M, K, B, N=20000, layout=0):
        op = g.StreamCorrelator(sysobj, N, M, B, K, shifts, N / 1e-3, ctx=vector_ctx)
        op.set_params(g.make_params(np.arange(K) % 32, 1.023e6, 1500.0, 0.0, 0.0, shape=(B, K)))
        if layout == 0:
            op(torch.zeros((M, B * N), device="cuda"), torch.zeros((M, B * N), device="cuda"))
        else:
            op(torch.zeros((M, B * N, 2), dtype=torch.int8 if layout == 3 else torch.int16, device="cuda"), None)
        i = vector_ctx.last_launch_info()
        return i["ant_tile"], i["channels_per_wg"]

    assert tile(4, 8, 256) == (4, 2) and tile(8, 4, 256) == (4, 2) and tile(12, 2, 512) == (4, 2)
    assert tile(4, 1, 1024) == (4, 1)                      # one channel: the one-wave-of-four tile
    assert tile(16, 4, 64) == (16, 4)                      # sixteen antennas: AW = 4, four channels per workgroup
    assert tile(4, 12, 1) == (4, 1)                        # one block: latency regime
    assert tile(4, 8, 256, layout=3) == (4, 1)             # int8 pairs
    assert tile(4, 8, 256, layout=2) == (4, 2)             # int16 pairs


def test_chip_edges_exact_in_the_two_by_two_tile(gat, two_by_two_ctx):
    """All-ones signal, zero carrier: integer sums, compared exactly with the oracle -- GPS L5 at 50 MHz with code phases at and
    next to chip edges and the table's wrap (the quads read "this chip and the next": index Lc - 1 is followed by chip 0), in
    every fill mode of the two-channel tile; and the adversarial ratio = 1/16 case on GPS L1 (every batch redone exactly)."""
    import torch
    g = gat
    ctx = two_by_two_ctx
    for system_name, N, fs, fc, L, taus in (
            ("GPSL5", 50000, 50e6, 10.23e6, 5, [[0.0, 10229.999999, 5115.5, 1e-9], [3.0 - 1e-12, 7.25, 10229.0, 10229.5]]),
            ("GPSL1", 8192, 1.023e6 * 16, 1.023e6, 3, [[0.0, 0.0625, 511.9375, 1022.9375], [512.0 - np.spacing(512.0), 1022.0 + 15 * 0.0625, 700.5, 1e-300]])):
        M, K, B = 4, 4, 2
        system = g.GNSSDICT[system_name](use_gpu=True)
        shifts = oracle.sample_shifts(L, fs, fc) if system_name == "GPSL5" else np.array([-8, 0, 8], dtype=np.int32)
        rng = np.random.default_rng(6)
        prm = oracle.make_params(np.broadcast_to(np.arange(K), (B, K)), fc * (1 + (rng.uniform(-3e-6, 3e-6, (B, K)) if system_name == "GPSL5" else 0.0)),
                                 np.zeros((B, K)), np.array(taus), np.zeros((B, K)))
        re = np.ones((M, B * N), dtype=np.float32)
        im = np.zeros_like(re)
        ref = oracle.correlate_f64(re, im, oracle.codes(system_name, 32), prm, fs, shifts, N=N)
        assert np.all(ref.imag == 0) and np.all(ref.real == np.rint(ref.real))
        op = g.StreamCorrelator(system, N, M, B, K, shifts, fs, ctx=ctx)
        op.set_params(g.make_params(prm["prn0"], prm["code_freq_hz"], prm["carrier_freq_hz"], prm["code_phase_chips"], prm["carrier_phase_cycles"]))
        d_re, d_im = torch.from_numpy(re).to(ctx.device), torch.from_numpy(im).to(ctx.device)
        ctx.set_option("dc_aw2", 1)
        for mode in FILL_MODES:
            for name, val in (("dc_quads", -1), ("dc_bits", 1), ("dc_seg", 0)):
                ctx.set_option(name, val)
            for name, val in mode.items():
                ctx.set_option(name, val)
            op(d_re, d_im)
            got = op.result()
            assert ctx.last_launch_info()["channels_per_wg"] == 2
            assert np.array_equal(got.real.astype(np.float64), ref.real) and np.all(got.imag == 0), (system_name, mode)


ONE_WAVE_SHAPES = [
    # system, N, M, L, K, B  -- short blocks of one- and two-antenna tiles: one wave per (block, channel, tile)
    ("GPSL1", 4000, 1, 3, 1, 40),    # configs[0] shape in a stream
    ("GPSL1", 2048, 2, 3, 3, 17),    # two antennas per wave, 3 channel groups
    ("GPSL1", 1001, 1, 5, 2, 9),     # ragged: 1001 is no multiple of the group size -> scalar path keeps four waves
    ("GPSL1", 1000, 1, 4, 1, 33),    # 4 taps; several segments per block (4 steps each)
    ("GPSL1", 6000, 2, 7, 2, 5),     # 7 taps
    ("GPSL1", 260, 1, 3, 1, 64),     # barely more than one step
]


@pytest.mark.parametrize("layout", [0, 1, 2, 3], ids=["planar", "interleaved", "i16", "i8"])
@pytest.mark.parametrize("shape", ONE_WAVE_SHAPES, ids=[f"N{s[1]}-M{s[2]}-L{s[3]}-K{s[4]}-B{s[5]}" for s in ONE_WAVE_SHAPES])
def test_one_wave_workgroups_match_the_oracle(gat, shape, layout):
    """Short blocks in a long stream run one wave per block (gat_dc.h, NW = 1).  The planner takes that path from 32
    groups per CU on; the option dc_one_wave_min = 1 (gat_set_option) takes it for these small cases too."""
    import torch
    ctx = gat.Context(torch.cuda.current_device())
    try:
        ctx.set_option("dc_one_wave_min", 1)
        ctx.set_matrix_core(gat.GAT_MC_VECTOR)
        system, N, M, L, K, B = shape
        case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=B)
        if layout >= 2:  # integer front-end samples: quantise the case itself so that the oracle sees the same numbers
            s = 100.0 if layout == 2 else 20.0
            case["re"] = np.rint(case["re"] * s).astype(np.float32)
            case["im"] = np.rint(case["im"] * s).astype(np.float32)
        got, info = run_case(gat, ctx, case, layout=layout)
        group = {0: 4, 1: 2, 2: 4, 3: 8}[layout]
        # (blocks are stored N apart here: a length that is no multiple of the load group also misaligns every later
        # block's start, which is what sends the case to the scalar-load kernel and its four-wave geometry)
        assert info["threads"] == (64 if N % group == 0 else 256), info
        check_close(got, oracle_result(case), what=f"{shape} {info}")
        # and the four-wave geometry on the same inputs: same chip edges, same sums up to summation order
        ctx.set_vector_tiling(1, 1, 1)
        ref, info4 = run_case(gat, ctx, case, layout=layout)
        assert info4["threads"] == 256, info4
        scale = np.abs(ref).max(axis=(2, 3), keepdims=True)
        assert np.max(np.abs(got - ref) / scale) < 2e-6
    finally:
        ctx.close()


def test_antenna_tile_whose_span_exceeds_a_descriptor(gat, vector_ctx):
    """A wave reaches its antennas through one buffer descriptor per plane (antenna = scalar offset, lanes past the
    block end get the offset 2^31): the host must tile fewer antennas per wave when (MT - 1) * ant_stride + N samples do
    not fit 2^31 bytes.  Four antennas 200 M samples apart (0.8 GB): two per wave still fit, four do not."""
    import torch
    shape = ("GPSL1", 6000, 4, 3, 2, 1)
    system, N, M, L, K, B = shape
    case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=B)
    dev = vector_ctx.device
    stride = 200_000_000
    re = torch.zeros((M - 1) * stride + N, dtype=torch.float32, device=dev)
    im = torch.zeros_like(re)
    for m in range(M):
        re[m * stride:m * stride + N] = torch.from_numpy(case["re"][m]).to(dev)
        im[m * stride:m * stride + N] = torch.from_numpy(case["im"][m]).to(dev)
    sysobj = gat.GNSSDICT[system](use_gpu=True)
    op = gat.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"], ctx=vector_ctx)
    p = case["prm"]
    op.set_params(gat.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"],
                                  p["carrier_phase_cycles"]))
    from gpuacceleratedtracking_amd import _lib
    desc = _lib.SignalDesc(re.data_ptr(), im.data_ptr(), gat.GAT_LAYOUT_PLANAR, M, N, stride, N, 0)
    op.launch(desc)
    got, info = op.result(), vector_ctx.last_launch_info()
    assert info["matrix_core"] == 0 and info["ant_tile"] == 2, info  # 3 * 0.8 GB + N > 2^31 > 1 * 0.8 GB + N
    check_close(got, oracle_result(case), what=f"wide antenna stride {info}")
    del re, im
    torch.cuda.empty_cache()


DEEP_SHAPES = [
    # system, N, M, L, K, B -- four antennas, <= 3 taps, one channel, no split: two register sets of samples per wave
    ("GPSL1", 20000, 4, 3, 1, 2100),   # configs[1] itself in a stream long enough not to be split (20 steps per block)
    ("GPSL1", 3000, 4, 3, 1, 7),       # 3 steps: the block's last group of two is padded with a zero step
    ("GPSL1", 2048, 4, 2, 1, 5),       # exactly 2 steps, 2 taps
    ("GPSL1", 4096, 4, 1, 1, 3),       # 4 steps, one tap
    ("GPSL1", 3000, 4, 3, 1, 9000),    # short blocks in a long stream: several blocks per workgroup, prefetch across blocks
    ("GPSL1", 1028, 4, 3, 1, 4),       # 2 steps, the second one nearly empty
]


@pytest.mark.parametrize("layout", [0, 1, 2, 3], ids=["planar", "interleaved", "i16", "i8"])
@pytest.mark.parametrize("shape", DEEP_SHAPES, ids=[f"N{s[1]}-L{s[3]}-B{s[5]}" for s in DEEP_SHAPES])
def test_two_sample_sets_match_the_oracle_and_one_set(gat, shape, layout):
    """The streaming regime of the four-antenna tile keeps two steps of samples in flight (gat_dc.h, D = 2): against
    the oracle (first and last blocks of long streams) and against the same launch with one set (option dc_depth = 1,
    gat_set_option): same chips, same order of summation -> bit-identical."""
    import torch
    system, N, M, L, K, B = shape
    if layout == 3 and N % 8:
        pytest.skip("int8 groups hold 8 samples")
    big = B > 100
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=min(B, 6))
    if layout >= 2:
        s = 100.0 if layout == 2 else 20.0
        case["re"] = np.rint(case["re"] * s).astype(np.float32)
        case["im"] = np.rint(case["im"] * s).astype(np.float32)
    nb = case["B"]
    if big:  # tile the small case's blocks over the long stream (parameters repeat with them)
        reps = (B + nb - 1) // nb
        for key in ("re", "im"):
            case[key] = np.tile(case[key].reshape(M, nb, N), (1, reps, 1))[:, :B].reshape(M, B * N)
        case["prm"] = np.tile(case["prm"], (reps, 1))[:B]
        case["B"] = B
    outs = []
    for depth in ("2", "1"):
        ctx = gat.Context(torch.cuda.current_device())
        try:
            ctx.set_option("dc_depth", int(depth))
            ctx.set_matrix_core(gat.GAT_MC_VECTOR)
            got, info = run_case(gat, ctx, case, layout=layout)
            steps = -(-N // (2048 if layout == 3 else 1024))  # a block of one step has nothing to prefetch
            want = int(depth) if steps >= 2 and layout <= 1 else 1  # float samples only (the planner's rule)
            assert info["prefetch_depth"] == want and info["splits"] == 1, info
            if big:
                assert info["blocks_per_wg"] > 1 or N >= 20000, info
            outs.append(got)
        finally:
            ctx.close()
    assert np.array_equal(outs[0], outs[1])
    small = dict(case)
    if big:  # oracle on the first blocks and the last ones
        for sel in (slice(0, nb), slice(B - nb, B)):
            sub = dict(case)
            sub["B"] = nb
            idx = np.arange(B)[sel]
            sub["re"] = case["re"].reshape(M, B, N)[:, idx].reshape(M, nb * N)
            sub["im"] = case["im"].reshape(M, B, N)[:, idx].reshape(M, nb * N)
            sub["prm"] = case["prm"][idx]
            check_close(outs[0][sel], oracle_result(sub), what=f"{shape} layout {layout} blocks {sel}")
    else:
        check_close(outs[0], oracle_result(small), what=f"{shape} layout {layout}")


RAGGED = [(2046, 4, "planar"), (2500, 4, "i8"), (4099, 4, "planar"), (2501, 2, "interleaved"), (4002, 1, "i16"), (4004, 1, "i8"),
          (5, 2, "i8"), (3, 4, "planar")]


@pytest.mark.parametrize("N,M,layout", RAGGED, ids=[f"N{n}-M{m}-{l}" for n, m, l in RAGGED])
def test_ragged_block_length_runs_the_vector_kernel(gat, N, M, layout):
    """The reference bounds every thread by num_samples (src/algorithms.jl:170): any block length works.  Here the 16-byte
    vector path needs every block of every antenna to START on a 16-byte boundary (strides padded to the load group), but
    the length itself may be anything -- N = 2046 (fs = 2 x 1.023 MHz), the reference's N = 2500 fixture as int8 pairs
    (2500 % 8 = 4), N = 4099: lanes behind the last whole group read zeros through the buffer range check and
    dc_tail_kernel adds the N % S samples behind it.  Same result as the oracle and as the zero-padded length; vec == 4;
    and no cliff: at most 1.3 x the time of the padded length on a stream of >= 128 MB (the tail is one more dependent
    launch, ~ 6 us of device time whatever the size; round 3 sent such lengths to the scalar-load kernel, 5-10 x slower)."""
    import torch
    g = gat
    case = make_case(321 + N, N=N, M=M, K=2, B=4, L=3)
    sysobj = g.GNSSDICT[case["system"]](use_gpu=True)
    dev = g.get_context().device
    prm = g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                        case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"])
    B = case["B"]
    spv = {"planar": 4, "interleaved": 2, "i16": 4, "i8": 8}[layout]
    lay = {"planar": g.GAT_LAYOUT_PLANAR, "interleaved": g.GAT_LAYOUT_INTERLEAVED, "i16": g.GAT_LAYOUT_INTERLEAVED_I16,
           "i8": g.GAT_LAYOUT_INTERLEAVED_I8}[layout]
    Np = (N + spv - 1) // spv * spv
    assert Np != N

    def stream(nblk, src_re, src_im):
        """blocks stored Np apart (aligned strides), zeros in the padding"""
        re = torch.zeros((M, nblk * Np), device=dev)
        im = torch.zeros((M, nblk * Np), device=dev)
        re.view(M, nblk, Np)[:, :, :N] = src_re
        im.view(M, nblk, Np)[:, :, :N] = src_im
        if layout == "planar":
            return (re, im), re, im
        if layout == "interleaved":
            x = torch.stack([re, im], dim=-1).contiguous()
            return (x,), re, im
        scale, lim, dt = (100.0, 32767, torch.int16) if layout == "i16" else (20.0, 127, torch.int8)
        x = torch.stack([torch.clamp(torch.round(re * scale), -lim - 1, lim), torch.clamp(torch.round(im * scale), -lim - 1, lim)],
                        dim=-1).to(dt).contiguous()
        return (x,), x[..., 0].float(), x[..., 1].float()

    keep, q_re, q_im = stream(B, torch.from_numpy(case["re"]).to(dev).view(M, B, N), torch.from_numpy(case["im"]).to(dev).view(M, B, N))
    ref_case = dict(case)  # what the kernel is given, block after block without the padding
    ref_case["re"] = q_re.view(M, B, Np)[:, :, :N].reshape(M, B * N).cpu().numpy()
    ref_case["im"] = q_im.view(M, B, Np)[:, :, :N].reshape(M, B * N).cpu().numpy()
    ref = oracle_result(ref_case)

    def desc_for(bufs, n, nblk):
        return g._lib.SignalDesc(bufs[0].data_ptr(), bufs[1].data_ptr() if layout == "planar" else None, lay, M, n, nblk * Np, Np, 0)

    out, info = {}, {}
    for name, n in (("ragged", N), ("padded", Np)):
        for flags in (0, g.GAT_FLAG_ATOMIC):
            op = g.StreamCorrelator(sysobj, n, M, B, 2, case["shifts"], case["fs"], flags=flags)
            op.set_params(prm)
            op.launch(desc_for(keep, n, B))
            out[name, flags] = op.result()
            info[name, flags] = op.ctx.last_launch_info()
    for key, i in info.items():
        assert i["vec"] == 4, (key, i)
    for key, got in out.items():
        check_close(got, ref, what=f"N={N} {layout} {key}")
    # B = 1, M = 1: strides that are never applied do not matter (a single-block call with any N takes the vector path)
    op = g.StreamCorrelator(sysobj, N, 1, 1, 2, case["shifts"], case["fs"])
    op.set_params(prm[:1])
    d1 = desc_for(keep, N, B)
    d1.num_ants, d1.ant_stride, d1.block_stride = 1, N, N
    op.launch(d1)
    assert op.ctx.last_launch_info()["vec"] == 4
    check_close(op.result(), ref[:1, :, :, :1], what=f"N={N} {layout} single block, one antenna")

    # ---- no cliff: a long stream of ragged blocks against the same stream told the padded length
    if N < 1000:
        return
    sample_bytes = {"planar": 8, "interleaved": 8, "i16": 4, "i8": 2}[layout]
    big = max(2048, -(-(128 << 20) // (Np * M * sample_bytes)) // B * B)
    rng = torch.Generator(device=dev).manual_seed(N)
    bufs, _, _ = stream(big, torch.randn((M, big, N), device=dev, generator=rng), torch.randn((M, big, N), device=dev, generator=rng))
    prm_big = np.tile(prm, (big // B, 1))
    times = {}
    for name, n in (("ragged", N), ("padded", Np)):
        op = g.StreamCorrelator(sysobj, n, M, big, 2, case["shifts"], case["fs"])
        op.set_params(prm_big)
        d = desc_for(bufs, n, big)
        for _ in range(30):
            op.launch(d)
        op.ctx.timer_start()
        for _ in range(40):
            op.launch(d)
        times[name] = op.ctx.timer_stop() / 40
    print(f"N={N} {layout} x {big} blocks: ragged {times['ragged'] * 1e3:.1f} us, padded {times['padded'] * 1e3:.1f} us per launch")
    assert times["ragged"] <= 1.3 * times["padded"], times


ALIGN_SHAPES = [
    # system, N, M, L, K, B -- block lengths whose byte size is no multiple of 128: blocks start 16 .. 112 bytes into a line
    ("GPSL1", 5004, 4, 3, 1, 9),     # planar: 20 016 B per block -> the start walks through all eight 16-byte offsets
    ("GPSL1", 5000, 16, 3, 4, 3),    # configs[3] shard shape (shortened): AW = 4, KT = 4, every other block 64 B off
    ("GPSL5", 6020, 4, 5, 3, 5),     # configs[2] family, three channel groups sharing a tile
    ("GPSL1", 2052, 4, 3, 1, 40),    # two steps per block, 40 blocks
    ("GPSL1", 24580, 4, 3, 1, 1),    # one block, split over workgroups: the base pointer itself is moved off the line
]


@pytest.mark.parametrize("layout", [0, 1, 2, 3], ids=["planar", "interleaved", "i16", "i8"])
@pytest.mark.parametrize("shape", ALIGN_SHAPES, ids=[f"{s[0]}-N{s[1]}-M{s[2]}-K{s[4]}-B{s[5]}" for s in ALIGN_SHAPES])
def test_blocks_that_start_off_a_cache_line(gat, shape, layout):
    """Where a block may start off a 128-byte line the workgroups walk it from the line's boundary (gat_dc.h: virtual block
    start, lanes in front of the real start read zeros): against the oracle, for every 16-byte offset of a block start within
    a line, with the walk from the sample itself (option dc_align = 0) beside it -- same chips, same samples, only the
    lane a sample lands in differs, so the two agree to summation order."""
    import torch
    g = gat
    system, N, M, L, K, B = shape
    group = {0: 4, 1: 2, 2: 4, 3: 8}[layout]
    if N % group:
        N -= N % group
    case = make_case(zlib.crc32(repr(shape).encode()) + layout, system=system, N=N, M=M, L=L, K=K, B=B)
    if layout >= 2:
        s = 100.0 if layout == 2 else 20.0
        case["re"] = np.rint(case["re"] * s).astype(np.float32)
        case["im"] = np.rint(case["im"] * s).astype(np.float32)
    ref = oracle_result(case)
    sysobj = g.GNSSDICT[system](use_gpu=True)
    prm = g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                        case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"])
    dev = torch.device("cuda", torch.cuda.current_device())
    lay = [g.GAT_LAYOUT_PLANAR, g.GAT_LAYOUT_INTERLEAVED, g.GAT_LAYOUT_INTERLEAVED_I16, g.GAT_LAYOUT_INTERLEAVED_I8][layout]
    sample_bytes = [4, 8, 4, 2][layout]
    outs = {}
    for lead in (0, 16 // sample_bytes * 3):  # the stream itself starts on a line / 48 bytes into one
        re = torch.zeros((M, lead + B * N), device=dev)
        im = torch.zeros((M, lead + B * N), device=dev)
        re[:, lead:] = torch.from_numpy(case["re"]).to(dev)
        im[:, lead:] = torch.from_numpy(case["im"]).to(dev)
        if layout == 0:
            bufs = (re, im)
            ptrs = (re.data_ptr() + lead * 4, im.data_ptr() + lead * 4)
        else:
            x = torch.stack([re, im], dim=-1)
            x = (x if layout == 1 else x.to(torch.int16 if layout == 2 else torch.int8)).contiguous()
            bufs = (x,)
            ptrs = (x.data_ptr() + lead * sample_bytes, None)
        desc = g._lib.SignalDesc(ptrs[0], ptrs[1], lay, M, N, lead + B * N, N, 0)
        for align in (1, 0):
            ctx = g.Context(torch.cuda.current_device())
            try:
                ctx.set_matrix_core(g.GAT_MC_VECTOR)
                ctx.set_option("dc_align", align)
                op = g.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"], ctx=ctx)
                op.set_params(prm)
                op.launch(desc)
                got = op.result()
                info = ctx.last_launch_info()
                assert info["vec"] == 4, info
                check_close(got, ref, what=f"{shape} layout {layout} lead {lead} align {align} {info}")
                outs[lead, align] = got
            finally:
                ctx.close()
        del bufs
    scale = np.abs(ref).max(axis=(2, 3), keepdims=True)
    for key, got in outs.items():
        assert np.max(np.abs(got - outs[0, 1]) / scale) < 2e-6, key
