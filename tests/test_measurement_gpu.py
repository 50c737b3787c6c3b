"""GPU tests of the measurement side of the C ABI (round 5): per-step lap timers, the read-only stream kernel behind
bench.py's `roofline.read_ceiling_GBps`, a BOX-TOLERANT performance guard for the headline shape (kernel time against the
in-run reader over the same bytes, not against a number measured on some other box) with the planner's choice of instance
for every BASELINE shape, and the fixed-point model of the texture unit's addressing against its numpy definition.

Reference regime of the guard: the reference times the call it is given and nothing else (`@benchmark CUDA.@sync
kernel_algorithm(...)`, src/benchmarks.jl:120-146); the statistics it keeps per call are src/benchmarks.jl:1-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def test_lap_timers_give_every_launch_its_own_interval(g):
    op, desc, sig, prm = g.build_stream("GPSL1", 20000, 4, 3, 1, 256)
    ctx = op.ctx
    for _ in range(5):
        op.launch(desc)
    ctx.timer_start()
    ctx.timer_lap()
    for _ in range(12):
        op.launch(desc)
        ctx.timer_lap()
    total = ctx.timer_stop()
    laps = ctx.timer_laps()
    assert laps.size == 12 and (laps > 0).all()
    assert abs(laps.sum() - total) <= 0.05 * total + 0.02  # the same events' span, up to the pair's own records
    assert ctx.timer_laps().size == 0  # forgotten once read
    # a capacity below the lap count truncates, never overruns
    for _ in range(4):
        ctx.timer_lap()
    assert ctx.timer_laps(2).size == 2


def test_read_stream_reports_a_plausible_rate(g):
    import torch
    buf = torch.zeros(512 << 20, dtype=torch.uint8, device="cuda")  # 512 MiB: beyond the 256 MB Infinity Cache
    ctx = g.get_context()
    best = 0.0
    for variant in (0, 1, 4, 9):
        ms = ctx.read_stream_ms(buf, buf.numel(), variant=variant, launches=5)
        assert ms.shape == (5,) and (ms > 0).all()
        best = max(best, buf.numel() / float(np.median(ms[1:])) / 1e6)
    print(f"read-only kernel over 512 MiB: {best:.0f} GB/s")
    assert 2000.0 < best < 8000.0  # an MI355X reads at 6-7 TB/s; the spec peak is 8
    with pytest.raises(g.GatError):
        ctx.read_stream_ms(buf, 24, variant=0, launches=1)


# which instance the planner picks for the BASELINE shapes (gat_launch_info) -- a planner change that sends one of them
# elsewhere has to show up here, whatever the box's clock is
EXPECTED = {
    # name: (build_stream args, expected launch-info fields)
    "configs[1]": (("GPSL1", 20000, 4, 3, 1, 4096), dict(matrix_core=0, threads=256, ant_tile=4, vec=4, channels_per_wg=1, splits=1, prefetch_depth=2)),
    "configs[0] shape": (("GPSL1", 4000, 1, 3, 1, 16384), dict(matrix_core=0, threads=64, ant_tile=1, vec=4, channels_per_wg=1, splits=1, prefetch_depth=2)),
    # (round 5: the two-channel 2 x 2 tile -- two waves of two antennas, two channels per workgroup)
    "configs[2]": (("GPSL5", 50000, 4, 5, 12, 256), dict(matrix_core=0, threads=256, ant_tile=4, vec=4, channels_per_wg=2, splits=1, prefetch_depth=1)),
    "configs[3] shard": (("GPSL1", 50000, 16, 3, 4, 512), dict(matrix_core=0, threads=256, ant_tile=16, vec=4, channels_per_wg=4, splits=1, prefetch_depth=1)),
    "configs[3] whole": (("GPSL1", 50000, 16, 3, 32, 128), dict(matrix_core=0, threads=256, ant_tile=16, vec=4, channels_per_wg=4, splits=1, prefetch_depth=1)),
    # (round 5: the split-bf16 kernel's <4 row tiles, 4 column tiles> instance, 3 column groups x 256 splits = 3 rounds of the chip)
    "configs[4]": (("GPSL1", 2000000, 64, 3, 64, 1), dict(matrix_core=2, threads=1024, ant_tile=64, workgroups=768, splits=256, bf16_terms=3), dict(block_seconds=20e-3)),
    "configs[4] from int16 pairs": (("GPSL1", 2000000, 64, 3, 64, 1), dict(matrix_core=2, threads=1024, ant_tile=64, workgroups=768, splits=256, bf16_terms=2),
                                    dict(block_seconds=20e-3, layout=2)),
}


@pytest.mark.parametrize("name", list(EXPECTED))
def test_planner_picks_the_expected_instance(g, name):
    args, want = EXPECTED[name][:2]
    op, desc, sig, prm = g.build_stream(*args, **(EXPECTED[name][2] if len(EXPECTED[name]) > 2 else {}))
    op.launch(desc)
    op.ctx.sync()
    info = op.ctx.last_launch_info()
    got = {k: info[k] for k in want}
    assert got == want, (name, info)


def test_headline_kernel_stays_within_reach_of_the_read_ceiling(g):
    """configs[1] as bench.py runs it (B = 4096: 2.6 GB), 64 settle + 30 timed launches: the kernel's median time per launch
    against what a kernel that ONLY reads reaches over the same stream's re plane in this process.  Today the correlator
    takes 1.03-1.06 x the reader's time per byte; 1.12 x is the alarm (a planner or kernel regression of ~6 %, on any box)."""
    N, M, L, K, B = 20000, 4, 3, 1, 4096
    op, desc, sig, prm = g.build_stream("GPSL1", N, M, L, K, B)
    ctx = op.ctx
    for _ in range(64):
        op.launch(desc)
    ctx.timer_lap()
    for _ in range(30):
        op.launch(desc)
        ctx.timer_lap()
    t_kernel = float(np.median(ctx.timer_laps())) * 1e-3
    alg = g.algorithmic_bytes(B, N, M, L, K, 8)
    nbytes = sig[0].numel() * sig[0].element_size()
    t_read = min(float(np.median(ctx.read_stream_ms(sig[0], nbytes, variant=v, launches=7)[1:])) for v in (0, 1, 2, 4, 8)) * 1e-3
    kernel_rate, read_rate = alg / t_kernel / 1e9, nbytes / t_read / 1e9
    print(f"configs[1]: kernel {t_kernel * 1e3:.4f} ms = {kernel_rate:.0f} GB/s algorithmic; reader {read_rate:.0f} GB/s; ratio {read_rate / kernel_rate:.3f}")
    assert kernel_rate * 1.12 >= read_rate, (kernel_rate, read_rate)
    assert kernel_rate >= 0.70 * 8000.0  # the north star's bar on any box this has run on (0.81-0.87)


@pytest.mark.parametrize("mode", [(0, -1), (0, 8), (20, -1), (23, 8), (16, 4)])
def test_texture_addressing_model_matches_its_definition(g, mode):
    """gat_gen_code_replica_texaddr against a numpy statement of the same steps (float32 normalised coordinate, wrap,
    truncation to a fixed-point coordinate, exact product, rounding to a fixed-point texel address, floor)."""
    import torch
    cb, tb = mode
    system = g.GPSL1(use_gpu=True)
    lc, fc = g.get_code_length(system), g.get_code_frequency(system)
    N = 32736  # fs = 32 x 1.023 MHz: every 32nd sample sits exactly on a chip edge -- where the models differ most
    fs = N / 1e-3
    count, first = N + 32, -16
    rep = torch.zeros(count, device="cuda")
    g.get_context().set_codes(system.codes)
    g.get_context().gen_code_replica(rep, count, 0, fc, fs, 0.25, first, texture_addressing=(cb, tb))
    i = np.arange(count, dtype=np.float64) + first
    p = (fc / fs) * i + 0.25
    u = (p / lc).astype(np.float32)
    w = (u - np.floor(u)).astype(np.float64)
    if cb > 0:
        w = np.floor(w * 2.0 ** cb) / 2.0 ** cb
    x = w * lc
    if tb >= 0:
        x = np.rint(x * 2.0 ** tb) / 2.0 ** tb
    idx = np.floor(x).astype(np.int64)
    idx = np.where(idx >= lc, idx - lc, np.maximum(idx, 0))
    want = system.codes[0][idx].astype(np.float32)
    assert np.array_equal(rep.cpu().numpy(), want)
