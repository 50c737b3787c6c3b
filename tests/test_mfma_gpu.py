"""GPU tests of the matrix-core kernels (antenna-rich shapes, M % 16 == 0) -- the split-bf16 kernel
(gat_mfma_bf16.hip, the default) and the f32-MFMA kernel (gat_mfma.hip): parity with the FP64 oracle,
agreement with the vector kernel on the same inputs, and that the planner picks them exactly for the
shapes they are meant for."""
import zlib

import numpy as np
import pytest

from tests.helpers import check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def run(g, case, matrix_core=1, flags=0):
    import torch
    sysobj = g.GNSSDICT[case["system"]](use_gpu=True)
    ctx = g.get_context()
    ctx.set_matrix_core(matrix_core)
    try:
        dev = ctx.device
        op = g.StreamCorrelator(sysobj, case["N"], case["M"], case["B"], case["K"], case["shifts"], case["fs"], flags=flags)
        p = case["prm"]
        op.set_params(g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"],
                                    p["carrier_phase_cycles"]))
        op(torch.from_numpy(case["re"]).to(dev), torch.from_numpy(case["im"]).to(dev))
        return op.result(), ctx.last_launch_info()
    finally:
        ctx.set_matrix_core(1)


def planner_auto_kind(M, K, L, layout="f32", N=None, B=1, num_cus=256):
    """Mirror of the planner's GAT_MC_AUTO rule for M % 16 == 0 (gat_planner.cpp, round 5): column tiles per workgroup by
    slot count x slot cost, then the split-bf16 kernel iff the live share of its slots reaches the layout's threshold
    (profiles/r05/mfma_planner_scan_*.txt; tap counts beyond three keep round 2's rule) AND the launch is long enough to
    give every workgroup 16 (int16 / int8) or 32 (float) steps (profiles/r05/mfma_blocks_scan_*.txt)."""
    tiles = (2 * L * K + 31) // 32
    rt = 4 if (M // 16) % 4 == 0 else 2 if (M // 16) % 2 == 0 else 1
    w2 = {"i8": (1.41, 1.22), "i16": (1.30, 1.13), "f32": (1.26, 1.235)}[layout][0 if rt == 4 else 1]
    n = 4 if tiles >= 4 else 2 if tiles >= 2 else 1
    if tiles >= 3:
        n = 2 if ((tiles + 1) // 2 * 2) * w2 < (tiles + 3) // 4 * 4 else 4
    slots = -(-tiles // n) * n
    util = 2 * L * K / (32 * slots * (1.0 if n == 4 else w2 if n == 2 else 1.8))
    util_min = 2.0 if rt == 1 else {"i16": (0.50, 0.53), "f32": (0.70, 0.90)}.get(layout, (0, 0))[0 if rt == 4 else 1]
    if 2 * L * K < 24:
        return 0
    if layout != "i8" and not (util >= util_min if L <= 3 else (M >= 32 and K >= 32 and M * K >= 2048)):
        return 0
    if N is not None:
        rt_i = 2 if (rt == 4 and n == 1) else rt
        T = min(32 * (4 // n), 128 // rt_i)
        steps = -(-N // T)
        groups = B * (M // (16 * rt_i)) * -(-tiles // n)
        cut = min(steps, max(1, -(-2 * num_cus // groups)))
        if -(-steps // cut) < (32 if layout == "f32" else 16):
            return 0
    return 2


GRID = [
    # system, N, M, L, K, B
    ("GPSL1", 50000, 16, 3, 4, 2),    # BASELINE config 4 per-GPU shape
    ("GPSL1", 20000, 64, 3, 12, 1),   # config-5-like: 64 antennas, 3 channel tiles
    ("GPSL1", 5000, 16, 3, 5, 3),     # exactly one full channel tile
    ("GPSL1", 5000, 16, 3, 6, 1),     # CT + 1 channels: second tile nearly empty
    ("GPSL1", 3004, 32, 3, 21, 1),    # 4 column tiles, nearly full: the split-bf16 kernel by default (round 5)
    ("GPSL5", 8192, 16, 5, 7, 2),     # L = 5 -> CT = 3
    ("GPSL1", 776, 16, 1, 16, 2, 0),  # L = 1: the split-bf16 tile does not fit, M x K is small for the f32 kernel -> auto: vector
    ("GPSL1", 260, 16, 16, 2, 1),     # L = 16: one channel per tile
    ("GPSL1", 100, 16, 3, 4, 4),      # shorter than one tile
    ("GPSL1", 200000, 16, 3, 4, 1),   # many steps, split over workgroups + finalize
    ("GPSL1", 9000, 32, 3, 7, 2),     # 2 row tiles per workgroup, 2 channel tiles
    ("GPSL1", 9000, 64, 3, 3, 1, 0),  # 4 row tiles, 18 of 32 columns: below the auto threshold (vector kernel faster)
    ("GPSL1", 70004, 48, 3, 10, 1),   # 3 antenna tiles of one row tile each, ragged last step
    ("GPSL5", 30000, 64, 5, 12, 1),   # L5: 12 channels x 5 taps = 4 channel tiles of CT = 3
    ("GPSL1", 12000, 64, 3, 32, 1),   # 6 column tiles in 3 workgroups of two; ONE short block: 3 steps per workgroup -> vector by default
    ("GPSL1", 8192, 64, 3, 32, 24),   # the same tiles in a launch of 24 blocks: 32 steps per workgroup -> split-bf16 by default
    ("GPSL1", 8192, 64, 3, 16, 64),   # 3 tiles in one workgroup of four (a dead tile): split-bf16 by default at 64 antennas ...
    ("GPSL1", 8192, 32, 3, 16, 64),   # ... not at 32
]


@pytest.mark.parametrize("cfg", GRID, ids=lambda c: f"{c[0]}-N{c[1]}-M{c[2]}-L{c[3]}-K{c[4]}-B{c[5]}")
def test_mfma_parity(g, cfg):
    system, N, M, L, K, B = cfg[:6]
    # GAT_MC_AUTO: the split-bf16 kernel where enough of its tile slots carry live columns (planner_auto_kind below)
    auto_kind = cfg[6] if len(cfg) > 6 else planner_auto_kind(M, K, L, "f32", N, B)
    forced = 2 if not (L == 1 and K == 16) else 1  # GAT_MC_BF16_SPLIT takes every shape whose tile fits
    fs = {"GPSL1": 8e6, "GPSL5": 25e6}[system] if L <= 5 else 2.5e6
    case = make_case(zlib.crc32(repr(cfg).encode()), system=system, N=N, M=M, L=L, K=K, B=B, fs=fs, if_hz=1.1e6)
    ref = oracle_result(case)
    auto, info = run(g, case)  # auto: the split-bf16 kernel where it is the fastest
    assert info["matrix_core"] == auto_kind, info
    check_close(auto, ref, what=f"auto (kind {auto_kind}) mfma {cfg}")
    got = auto
    if forced == 2:
        got, info = run(g, case, matrix_core=g.GAT_MC_BF16_SPLIT)
        assert info["matrix_core"] == 2, info
        check_close(got, ref, what=f"split-bf16 mfma {cfg}")
    f32, info_f = run(g, case, matrix_core=g.GAT_MC_F32)
    assert info_f["matrix_core"] == 1, info_f
    check_close(f32, ref, what=f"f32 mfma {cfg}")
    vec, info_v = run(g, case, matrix_core=g.GAT_MC_VECTOR)
    assert info_v["matrix_core"] == 0
    check_close(vec, ref, what=f"vector {cfg}")
    # the kernels agree far inside the tolerance (f32 products / 8 of the 9 split products, different sum order)
    assert np.abs(f32 - vec).max() <= 3e-6 * np.abs(ref).max()
    assert np.abs(got - vec).max() <= 3e-6 * np.abs(ref).max()


def test_mfma_unsorted_taps_atomic_and_determinism(g):
    case = make_case(99, N=30000, M=16, L=3, K=5, B=1, fs=10e6)
    case["shifts"] = np.array([4, -4, 0, 9, -120], dtype=np.int32)
    case["L"] = 5
    ref = oracle_result(case)
    for mode, kind in ((g.GAT_MC_BF16_SPLIT, 2), (g.GAT_MC_F32, 1)):
        got, info = run(g, case, matrix_core=mode)
        assert info["matrix_core"] == kind and info["splits"] > 1, info
        check_close(got, ref, what="unsorted taps")
        got2, _ = run(g, case, matrix_core=mode)
        assert np.array_equal(got.view(np.float32), got2.view(np.float32))
        gota, _ = run(g, case, matrix_core=mode, flags=g.GAT_FLAG_ATOMIC)
        check_close(gota, ref, what="atomic")


def test_planner_keeps_vector_kernel_for_other_shapes(g):
    for (M, K, L) in ((4, 1, 3), (16, 1, 3), (12, 8, 3), (16, 4, 17), (16, 32, 3), (32, 8, 3), (64, 4, 3)):
        case = make_case(7, N=4000, M=M, L=L, K=K, B=1)
        _, info = run(g, case)
        assert info["matrix_core"] == 0, (M, K, L, info)


@pytest.mark.parametrize("fmt", ["interleaved", "int16", "int8"])
@pytest.mark.parametrize("cfg", [("GPSL1", 20000, 16, 3, 4, 2), ("GPSL1", 6000, 64, 3, 7, 1), ("GPSL5", 12000, 32, 5, 3, 1)],
                         ids=lambda c: f"{c[0]}-N{c[1]}-M{c[2]}-L{c[3]}-K{c[4]}-B{c[5]}")
def test_split_bf16_ingest_formats(g, cfg, fmt):
    """The split-bf16 kernel reads interleaved ComplexF32 / int16 / int8 pairs directly (an ADC delivers these; an
    int8 or int16 sample is exact in one or two bf16 terms): parity with the oracle on the exactly-converted integers
    and agreement with the vector kernel on the same device buffer."""
    import torch
    system, N, M, L, K, B = cfg
    case = make_case(zlib.crc32(repr((cfg, fmt)).encode()), system=system, N=N, M=M, L=L, K=K, B=B, fs=10e6, if_hz=2e5)
    if fmt == "interleaved":
        x = np.stack([case["re"], case["im"]], axis=-1)  # [M, B*N, 2] float32
    else:
        dtype, amp = (np.int16, 3000.0 / K) if fmt == "int16" else (np.int8, 25.0 / K)
        lim = np.iinfo(dtype)
        q_re = np.clip(np.rint(case["re"] * amp), lim.min, lim.max).astype(dtype)
        q_im = np.clip(np.rint(case["im"] * amp), lim.min, lim.max).astype(dtype)
        case["re"], case["im"] = q_re.astype(np.float32), q_im.astype(np.float32)  # what the oracle sees (exact)
        x = np.stack([q_re, q_im], axis=-1)
    ref = oracle_result(case)
    ctx = g.get_context()
    sysobj = g.GNSSDICT[system](use_gpu=True)
    p = case["prm"]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    xd = torch.from_numpy(x).to(ctx.device)
    res = {}
    try:
        for mode, kind in ((g.GAT_MC_BF16_SPLIT, 2), (g.GAT_MC_VECTOR, 0)):
            ctx.set_matrix_core(mode)
            op = g.StreamCorrelator(sysobj, N, M, B, K, case["shifts"], case["fs"])
            op.set_params(prm)
            op(xd, None)
            assert ctx.last_launch_info()["matrix_core"] == kind, (mode, ctx.last_launch_info())
            res[kind] = op.result()
            check_close(res[kind], ref, what=f"{fmt} {cfg} kernel {kind}")
    finally:
        ctx.set_matrix_core(1)
    assert np.abs(res[2] - res[0]).max() <= 3e-6 * np.abs(ref).max()


def test_planner_fallbacks_of_the_split_bf16_kernel(g):
    """Shapes the split-bf16 kernel does not take run on the f32-MFMA or the vector kernel -- same results."""
    # odd sample count: no 16-byte loads -> vector kernel
    case = make_case(11, N=4002, M=16, L=3, K=5, B=1)
    got, info = run(g, case)
    assert info["matrix_core"] == 0, info
    check_close(got, oracle_result(case), what="N % 4 != 0")
    # two taps: 8 channels per f32 tile; 20 channels pack flat into 80 columns = 3 split-bf16 tiles
    case = make_case(12, N=6000, M=16, L=2, K=20, B=2, fs=4e6)
    ref = oracle_result(case)
    for mode, kind in ((g.GAT_MC_BF16_SPLIT, 2), (g.GAT_MC_F32, 1), (g.GAT_MC_VECTOR, 0), (g.GAT_MC_AUTO, 0)):
        got, info = run(g, case, matrix_core=mode)
        assert info["matrix_core"] == kind, (mode, info)
        check_close(got, ref, what=f"L=2 K=20 mode {mode}")


def test_non_pm1_code_table_takes_the_f32_kernel(g):
    """A caller-supplied table with blanked (0) chips has no sign-bit form: the split-bf16 kernel must not take it (auto
    selection falls to the vector kernel at this size), the f32-MFMA kernel (which multiplies by the chip value) can --
    every mode matches the oracle."""
    import torch
    case = make_case(21, N=8000, M=16, L=3, K=5, B=1, fs=5e6)
    codes = case["codes"].copy()
    codes[:, ::7] = 0  # every 7th chip blanked
    case["codes"] = codes
    ref = oracle_result(case)
    ctx = g.get_context()
    sysobj = g.GPSL1(codes=codes, code_frequency=case["fc"])
    try:
        for mode, kind in ((g.GAT_MC_AUTO, 0), (g.GAT_MC_F32, 1), (g.GAT_MC_BF16_SPLIT, 0), (g.GAT_MC_VECTOR, 0)):
            ctx.set_matrix_core(mode)
            op = g.StreamCorrelator(sysobj, case["N"], case["M"], case["B"], case["K"], case["shifts"], case["fs"])
            p = case["prm"]
            op.set_params(g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"],
                                        p["carrier_phase_cycles"]))
            op(torch.from_numpy(case["re"]).to(ctx.device), torch.from_numpy(case["im"]).to(ctx.device))
            assert ctx.last_launch_info()["matrix_core"] == kind, (mode, ctx.last_launch_info())
            check_close(op.result(), ref, what=f"blanked chips, mode {mode}")
    finally:
        ctx.set_matrix_core(1)
        g.StreamCorrelator(g.GPSL1(), 64, 1, 1, 1, case["shifts"], case["fs"])  # restore the standard table


def test_mfma_bad_prn_poisons_output(g):
    import torch
    system = g.GPSL1()
    N, M, K = 4096, 16, 4
    ctx = g.get_context()
    re = torch.ones((M, N), device=ctx.device)
    im = torch.zeros_like(re)
    op = g.StreamCorrelator(system, N, M, 1, K, np.array([-1, 0, 1], dtype=np.int32), 4e6)
    prm = g.make_params(np.arange(K), 1.023e6, 0.0, 0.0, 0.0, shape=(1, K))
    op.set_params(prm)
    bad = prm.copy()
    bad["prn"][0, 2] = 77  # bypass set_params' host check: write the device copy directly
    op.params_dev = ctx.params_to_device(bad)
    op._prepared = None
    try:
        for mode, kind in ((g.GAT_MC_BF16_SPLIT, 2), (g.GAT_MC_F32, 1)):
            ctx.set_matrix_core(mode)
            op(re, im)
            out = op.result()
            assert ctx.last_launch_info()["matrix_core"] == kind
            assert np.isnan(out[0, 2].view(np.float32)).all() and np.isfinite(out[0, [0, 1, 3]].view(np.float32)).all()
    finally:
        ctx.set_matrix_core(1)


def test_split_bf16_extreme_dynamic_range(g):
    """hi + mid + lo carries all 24 mantissa bits whatever the scale: antennas with gains from 1e-12 to
    1e+12 next to each other (the matrix instruction never mixes rows) keep the f32-level agreement."""
    case = make_case(5, N=12000, M=16, L=3, K=5, B=1, fs=6e6)
    gains = (10.0 ** np.linspace(-12, 12, 16)).astype(np.float32)
    case["re"] = (case["re"].reshape(16, -1) * gains[:, None]).reshape(case["re"].shape)
    case["im"] = (case["im"].reshape(16, -1) * gains[:, None]).reshape(case["im"].shape)
    ref = oracle_result(case)
    got, info = run(g, case, matrix_core=g.GAT_MC_BF16_SPLIT)
    assert info["matrix_core"] == 2
    vec, _ = run(g, case, matrix_core=g.GAT_MC_VECTOR)
    scale = np.abs(ref).max(axis=(2,), keepdims=True)  # per (block, channel, antenna): max over taps
    assert (np.abs(got - ref) / scale).max() <= 1e-5
    assert (np.abs(vec - ref) / scale).max() <= 1e-5


# (M, K, L, N): one shape per instance <row tiles, channel tiles> of the split-bf16 kernel -- rt = M / 16 capped at 4 (2 when
# there is a single column tile), nct = column tiles (2 L K / 32) capped at 4
TWO_TERM_SHAPES = [(16, 2, 3, 6000), (16, 6, 3, 6000), (16, 16, 3, 4100), (32, 3, 3, 6000), (32, 8, 3, 4100), (32, 16, 3, 3000),
                   (64, 2, 3, 3000), (64, 9, 3, 3000), (64, 32, 3, 2600), (64, 12, 5, 2600)]


@pytest.mark.parametrize("bits", [16, 8, 3], ids=lambda b: f"{b}bit")
@pytest.mark.parametrize("shape", TWO_TERM_SHAPES, ids=lambda s: f"M{s[0]}-K{s[1]}-L{s[2]}-N{s[3]}")
def test_int16_two_term_split(g, shape, bits):
    """int16 samples on the split-bf16 kernel (round 5): v = a + b with a = v rounded to bf16's 8 significant bits and
    b = v - a, both exact in bf16 -- 5 products per sample instead of the float path's 8.  FULL-SCALE data, and the
    output of an 8-bit or 3-bit converter in the same container (there a cut at a FIXED bit would leave the whole sample
    in the term that meets only two of the carrier's three terms), with the corner values (-32768, 32767, -1, -256, 255,
    0) placed at the tile edges: parity with the FP64 oracle, agreement with the three-term path (option mc_i16_terms = 3)
    on the same device buffer far inside the tolerance, and the launch info names the split that ran."""
    import torch
    M, K, L, N = shape
    system = "GPSL5" if L == 5 else "GPSL1"
    case = make_case(zlib.crc32(repr(shape).encode()), system=system, N=N, M=M, L=L, K=K, B=2, fs=10e6, if_hz=2e5)
    rng = np.random.default_rng(N + M)
    q = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(2,) + case["re"].shape, dtype=np.int64).astype(np.int16)
    corners = np.array([-32768, 32767, -1, -256, 255, 0, 256, -255, -257, 1], dtype=np.int16)
    for comp in range(2):
        flat = q[comp].reshape(M, -1)
        flat[:, :10] = corners
        flat[:, 27:37] = corners[::-1]
        flat[:, -10:] = corners
        flat[M - 1, 60:70] = -32768
    case["re"], case["im"] = q[0].astype(np.float32), q[1].astype(np.float32)
    ref = oracle_result(case)
    ctx = g.get_context()
    sysobj = g.GNSSDICT[system](use_gpu=True)
    p = case["prm"]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    xd = torch.from_numpy(np.stack([q[0], q[1]], axis=-1)).to(ctx.device)
    res = {}
    try:
        ctx.set_matrix_core(g.GAT_MC_BF16_SPLIT)
        for terms in (2, 3):
            ctx.set_option("mc_i16_terms", terms)
            op = g.StreamCorrelator(sysobj, N, M, 2, K, case["shifts"], case["fs"])
            op.set_params(prm)
            op(xd, None)
            info = ctx.last_launch_info()
            assert info["matrix_core"] == 2 and info["bf16_terms"] == terms, info
            res[terms] = op.result()
            check_close(res[terms], ref, what=f"int16, {terms} bf16 terms per sample, {shape}")
    finally:
        ctx.set_option("mc_i16_terms", 2)
        ctx.set_matrix_core(1)
    assert np.abs(res[2] - res[3]).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize("shape", [(32, 8, 64), (32, 8, 4), (32, 4, 64), (16, 16, 64), (48, 16, 64), (64, 12, 32), (64, 12, 1), (64, 4, 32)],
                         ids=lambda s: f"M{s[0]}-K{s[1]}-B{s[2]}")
def test_planner_auto_rule_for_int16_samples(g, shape):
    """int16 pairs need 5/8 of the float path's MFMAs: GAT_MC_AUTO hands them to the split-bf16 kernel from two column tiles
    on when the antennas fill two or four row tiles (scripts/r05_i16_planner_scan.sh), never at one row tile or one column
    tile, and never when the launch is too short to give every workgroup 16 steps (one block, four blocks) -- and
    whichever kernel runs, the result is the oracle's."""
    import torch
    M, K, B = shape
    N, L = 16384, 3
    case = make_case(M * 131 + K, N=N, M=M, L=L, K=K, B=B, fs=8e6, if_hz=1e5)
    q = np.clip(np.rint(np.stack([case["re"], case["im"]], axis=-1) * (2000.0 / K)), -32768, 32767).astype(np.int16)
    case["re"], case["im"] = q[..., 0].astype(np.float32), q[..., 1].astype(np.float32)
    ctx = g.get_context()
    ctx.set_matrix_core(g.GAT_MC_AUTO)
    op = g.StreamCorrelator(g.GPSL1(use_gpu=True), N, M, B, K, case["shifts"], case["fs"])
    p = case["prm"]
    op.set_params(g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"]))
    op(torch.from_numpy(q).to(ctx.device), None)
    info = ctx.last_launch_info()
    assert info["matrix_core"] == planner_auto_kind(M, K, L, "i16", N, B), info
    assert info["bf16_terms"] == (2 if info["matrix_core"] == 2 else 0)
    check_close(op.result(), oracle_result(case), what=f"int16 auto {shape}")


# every instance <row tiles, column tiles> of the split-bf16 kernel: (M, K) that the planner maps onto it when mc_nct asks for
# that many column tiles (3 taps: 6 columns per channel)
INSTANCES = {(1, 1): (16, 5), (1, 2): (16, 10), (1, 4): (48, 21), (2, 1): (32, 5), (2, 2): (32, 10), (2, 4): (32, 21), (4, 2): (64, 10), (4, 4): (64, 21)}
INSTANCE_LAYOUTS = ["planar", "interleaved", "int16", "int16-three-terms", "int8"]


@pytest.mark.parametrize("layout", INSTANCE_LAYOUTS)
@pytest.mark.parametrize("inst", list(INSTANCES), ids=lambda i: f"rt{i[0]}-nct{i[1]}")
def test_every_instance_through_the_pipelined_step_loop(g, inst, layout):
    """All 40 instances of the split-bf16 kernel with ~13 steps per workgroup: the consumers' fragment fetches are pipelined
    across the step barrier from (fetch depth + 1) steps on (shorter launches -- most of the other cases here -- run the
    plain loop), and those fetches are inline-assembly LDS reads whose registers the compiler must not move before the
    inline-assembly wait: something only the result can tell, per instance (round 5: a select on the window position of
    the chip-sign ring, compiled into a branch, made the allocator copy fragments in flight in two of the 40 instances
    while the other 38 passed).  Parity with the oracle at 1e-5."""
    import torch
    rt, nct = inst
    M, K = INSTANCES[inst]
    T = min(32 * (4 // nct), 128 // rt)
    B, L = 16, 3
    N = 13 * 32 * T // (M // (16 * rt))  # ~13 steps for each of the ~512 workgroups the launch is cut into
    N -= N % 8
    case = make_case(zlib.crc32(repr((inst, layout)).encode()), N=N, M=M, L=L, K=K, B=B, fs=8e6, if_hz=1e5)
    ctx = g.get_context()
    if layout in ("planar", "interleaved"):
        if layout == "planar":
            sig = (torch.from_numpy(case["re"]).to(ctx.device), torch.from_numpy(case["im"]).to(ctx.device))
        else:
            sig = (torch.from_numpy(np.stack([case["re"], case["im"]], axis=-1)).to(ctx.device), None)
    else:
        dt, amp = (np.int8, 25.0 / K) if layout == "int8" else (np.int16, 6000.0 / K)
        lim = np.iinfo(dt)
        q = np.clip(np.rint(np.stack([case["re"], case["im"]], axis=-1) * amp), lim.min, lim.max).astype(dt)
        case["re"], case["im"] = q[..., 0].astype(np.float32), q[..., 1].astype(np.float32)
        sig = (torch.from_numpy(q).to(ctx.device), None)
    ref = oracle_result(case)
    p = case["prm"]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    try:
        ctx.set_matrix_core(g.GAT_MC_BF16_SPLIT)
        ctx.set_option("mc_nct", nct)
        ctx.set_option("mc_i16_terms", 3 if layout == "int16-three-terms" else 2)
        op = g.StreamCorrelator(g.GPSL1(use_gpu=True), N, M, B, K, case["shifts"], case["fs"])
        op.set_params(prm)
        op(*sig)
        info = ctx.last_launch_info()
        assert info["matrix_core"] == 2 and info["ant_tile"] == 16 * rt and info["threads"] in (768, 1024), info
        steps_each = -(-(-(-N // T)) // info["splits"])
        assert 8 <= steps_each <= 40, (steps_each, info)
        check_close(op.result(), ref, what=f"instance <{rt}, {nct}> {layout}: {steps_each} steps per workgroup")
    finally:
        ctx.set_option("mc_nct", 0)
        ctx.set_option("mc_i16_terms", 2)
        ctx.set_matrix_core(1)
