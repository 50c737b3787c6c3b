"""Resident correlator (include/gat.h gat_resident_*; kernel gat_resident.h): single-block calls rung into a kernel that
stays on the device -- parity with the FP64 oracle through every path of the doorbell protocol (one workgroup, several
workgroups polling the host's doorbell or a forwarded one, the host's second stage over their result lines, several
channels, block offsets), visibility of a signal rewritten between calls, and the kernel's bounded lifetime (idle exit,
call budget, park, restart).  Run with -m gpu."""
import time

import numpy as np
import pytest

from tests.helpers import check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gat():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def _params(g, case, b):
    p = case["prm"][b]
    return g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])


def _open(g, case, layout=0, **config):
    """The case's B blocks in one device buffer; the correlator is opened on block 0, block b is call offset b * N."""
    import torch
    ctx = g.get_context()
    ctx.set_codes(case["codes"])
    dev = ctx.device
    N, M, B = case["N"], case["M"], case["B"]
    re = torch.from_numpy(case["re"]).to(dev)
    im = torch.from_numpy(case["im"]).to(dev)
    if layout == 0:
        keep = (re, im)
        desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, B * N, N, 0)
    else:
        x = torch.stack([re, im], dim=-1).contiguous()
        keep = (x,)
        desc = g._lib.SignalDesc(x.data_ptr(), None, g.GAT_LAYOUT_INTERLEAVED, M, N, B * N, N, 0)
    torch.cuda.synchronize()
    try:
        res = ctx.open_resident(desc, case["K"], case["shifts"], case["fs"], **config)
    except g.GatError as e:
        if config.get("doorbell") == 2 and e.status == 4:  # GAT_ERR_UNSUPPORTED: the host cannot write device memory here
            pytest.skip("no large BAR: the doorbell cannot live in device memory on this box")
        raise
    res._keep = keep
    return ctx, res


@pytest.mark.parametrize("N,M,L,K,layout,max_wgs,pollers", [
    (2048, 4, 3, 1, 0, 1, 0),     # ONE workgroup: the reference grid's smallest point
    (2048, 4, 3, 1, 0, 0, 0),     # the same split over workgroups that all poll the host's doorbell
    (2048, 1, 7, 1, 0, 0, 0),
    (4096, 4, 3, 1, 1, 0, 0),     # ComplexF32 pairs
    (4096, 4, 8, 1, 0, 0, 1),     # 64 sums per workgroup (five result lines); the master forwards the ring
    (32768, 4, 3, 1, 0, 0, 0),    # 33 workgroups: forwarded doorbell
    (32768, 4, 3, 1, 0, 0, 64),   # ... and all 33 polling the host
    (32768, 1, 3, 1, 1, 8, 0),
    (16384, 16, 3, 1, 0, 16, 0),  # four antenna tiles x splits
    (8192, 2, 5, 3, 0, 0, 0),     # three channels: three doorbell lines, channel workgroups
    (2500 - 2500 % 4, 3, 3, 4, 0, 0, 0),
    (262144, 4, 3, 1, 0, 0, 0),   # the grid's largest point
    (262144, 1, 7, 1, 0, 0, 0),   # ... with seven taps half a chip apart: a tap span of 768 samples in one launch
    (8192, 2, 3, 7, 0, 0, 64),    # seven channels: two 256-byte doorbell groups, every workgroup polling the host
    (20000, 4, 3, 12, 0, 0, 0),   # a whole constellation in one call: twelve channels, forwarded doorbell (72 workgroups)
    (4096, 1, 3, 16, 1, 16, 0),   # sixteen channels, one workgroup each
    (2052, 4, 3, 1, 0, 0, 0),     # blocks that start 16 / 32 / 48 ... bytes into a 128-byte line: walked from the line, trimmed per call
    (20004, 3, 5, 2, 0, 0, 0),
    (6146, 2, 3, 1, 1, 0, 0),     # ComplexF32 pairs, block starts 16 bytes off the line
])
@pytest.mark.parametrize("bell", [2, 1])  # the doorbell in device memory behind the PCIe BAR | in pinned host memory (polled directly or forwarded)
def test_resident_matches_oracle(gat, N, M, L, K, layout, max_wgs, pollers, bell):
    g = gat
    case = make_case(900 + N % 97 + M + K, N=N, M=M, L=L, K=K, B=3)
    ref = oracle_result(case)
    ctx, res = _open(g, case, layout, max_workgroups=max_wgs, host_pollers=pollers, idle_us=200000, doorbell=bell)
    try:
        info = res.info()
        assert info["running"] == 1 and info["launches"] == 1
        for rep in range(2):  # every block twice: the second round runs on staged chip tables
            for b in (0, 2, 1):
                re, im = res.correlate(_params(g, case, b), block_offset=b * N)
                got = (re + 1j * im)[None]  # [1, K, L, M]
                assert np.isfinite(got.view(np.float32)).all()
                check_close(got, ref[b:b + 1], what=f"resident N={N} M={M} K={K} block {b} rep {rep}")
        assert res.info()["calls"] == 6 and res.info()["launches"] == 1
    finally:
        res.close()


@pytest.mark.parametrize("dtype,amp,layout", [(np.int16, 3000.0, 2), (np.int8, 25.0, 3)])
@pytest.mark.parametrize("N,M,L,K", [(4096, 4, 3, 1), (32768, 2, 5, 2)])
def test_resident_integer_samples(gat, N, M, L, K, dtype, amp, layout):
    """int16 / int8 {re, im} pairs as ADC front-ends deliver them: the oracle sees the exactly-converted integers."""
    import torch
    g = gat
    case = make_case(300 + N % 89 + M, N=N, M=M, L=L, K=K, B=2)
    lim = np.iinfo(dtype)
    q_re = np.clip(np.rint(case["re"] * amp / K), lim.min, lim.max).astype(dtype)
    q_im = np.clip(np.rint(case["im"] * amp / K), lim.min, lim.max).astype(dtype)
    case["re"], case["im"] = q_re.astype(np.float32), q_im.astype(np.float32)
    ref = oracle_result(case)
    ctx = g.get_context()
    ctx.set_codes(case["codes"])
    x = torch.from_numpy(np.stack([q_re, q_im], axis=-1)).to(ctx.device)  # [M, B * N, 2]
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(x.data_ptr(), None, layout, M, N, 2 * N, N, 0)
    with ctx.open_resident(desc, K, case["shifts"], case["fs"], idle_us=200000) as res:
        for b in (1, 0, 1):
            re, im = res.correlate(_params(g, case, b), block_offset=b * N)
            check_close((re + 1j * im)[None], ref[b:b + 1], what=f"{dtype.__name__} block {b}")


def test_resident_one_signal_per_channel(gat):
    """chan_stride != 0 (the reference's _3d_4431! form, src/algorithms.jl:668): channel k correlates ITS signal."""
    import torch
    g = gat
    N, M, L, K = 8192, 2, 3, 3
    cases = [make_case(40 + k, N=N, M=M, L=L, K=1, B=1) for k in range(K)]
    ctx = g.get_context()
    ctx.set_codes(cases[0]["codes"])
    re = torch.from_numpy(np.stack([c["re"] for c in cases])).to(ctx.device)  # [K, M, N]
    im = torch.from_numpy(np.stack([c["im"] for c in cases])).to(ctx.device)
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, M * N)
    prm = np.concatenate([_params(g, c, 0) for c in cases])
    with ctx.open_resident(desc, K, cases[0]["shifts"], cases[0]["fs"], idle_us=200000) as res:
        r, i = res.correlate(prm)
        for k, c in enumerate(cases):
            check_close((r[k:k + 1] + 1j * i[k:k + 1])[None], oracle_result(c), what=f"channel {k} on its own signal")


def test_resident_calls_are_bit_identical_and_agree_with_the_ordinary_call(gat):
    g = gat
    import torch
    case = make_case(77, N=16384, M=4, L=3, K=2, B=1)
    ctx, res = _open(g, case)
    try:
        prm = _params(g, case, 0)
        first = tuple(a.copy() for a in res.correlate(prm))
        for _ in range(20):
            re, im = res.correlate(prm)
            assert np.array_equal(re, first[0]) and np.array_equal(im, first[1])
        # the ordinary call on the same buffers (different split plan: summation order only)
        dev = ctx.device
        o_re = torch.zeros((1, 2, 3, 4), device=dev)
        o_im = torch.zeros_like(o_re)
        desc = res._desc
        ctx.downconvert_and_correlate(desc, prm, 1, 2, case["shifts"], case["fs"], o_re, o_im)
        torch.cuda.synchronize()
        ordinary = (o_re.cpu().numpy() + 1j * o_im.cpu().numpy())
        check_close((first[0] + 1j * first[1])[None], ordinary, what="resident vs ordinary call")
    finally:
        res.close()


def test_resident_sees_a_signal_rewritten_between_calls(gat):
    """The kernel that serves call n+1 is the one that read the buffer for call n: nothing of the old samples may come
    from its caches (its sample loads are system-scope loads) -- whoever rewrote the buffer: a copy from the host, a copy
    from another device buffer, a kernel that works in place."""
    g = gat
    import torch
    a = make_case(11, N=8192, M=4, L=3, K=1, B=1)
    b = make_case(12, N=8192, M=4, L=3, K=1, B=1)
    want = {id(a): oracle_result(a), id(b): oracle_result(b)}
    ctx, res = _open(g, a, idle_us=500000, life_ms=5000)
    try:
        re_t, im_t = res._keep
        dev = {id(c): (torch.from_numpy(c["re"]).to(re_t.device), torch.from_numpy(c["im"]).to(re_t.device)) for c in (a, b)}
        torch.cuda.current_stream().synchronize()
        for rnd in range(18):
            cur = a if rnd % 2 == 0 else b
            how = rnd % 3
            if how == 0:  # host -> device copy
                re_t.copy_(torch.from_numpy(cur["re"]))
                im_t.copy_(torch.from_numpy(cur["im"]))
            elif how == 1:  # device -> device copy
                re_t.copy_(dev[id(cur)][0])
                im_t.copy_(dev[id(cur)][1])
            else:  # a kernel that writes the buffer in place
                re_t.mul_(0.0).add_(dev[id(cur)][0])
                im_t.mul_(0.0).add_(dev[id(cur)][1])
            torch.cuda.current_stream().synchronize()  # (a device-wide wait would sit out the kernel's idle limit)
            re, im = res.correlate(_params(g, cur, 0))
            check_close((re + 1j * im)[None], want[id(cur)], what=f"round {rnd}")
        assert res.info()["launches"] == 1  # one kernel served all of them
    finally:
        res.close()


def test_resident_lifetime_is_bounded_on_the_device(gat):
    g = gat
    case = make_case(5, N=4096, M=4, L=3, K=1, B=1)
    ref = oracle_result(case)
    ctx, res = _open(g, case, idle_us=3000, life_ms=200, max_calls=7)
    try:
        prm = _params(g, case, 0)
        re, im = res.correlate(prm)
        check_close((re + 1j * im)[None], ref)
        # idle: the kernel leaves by itself
        time.sleep(0.05)
        i = res.info()
        assert i["running"] == 0 and i["last_exit"] == 2, i
        re, im = res.correlate(prm)  # starts it again
        check_close((re + 1j * im)[None], ref)
        assert res.info()["launches"] == 2
        # call budget: 7 calls per kernel
        for _ in range(20):
            re, im = res.correlate(prm)
        check_close((re + 1j * im)[None], ref)
        i = res.info()
        assert i["launches"] >= 4 and i["calls"] == 22, i
        # lifetime: calls every millisecond keep it from idling; it still leaves after life_ms
        res.park()
        assert res.info()["running"] == 0 and res.info()["last_exit"] in (1, 2, 4)
        n0 = res.info()["launches"]
        t0 = time.time()
        while time.time() - t0 < 0.5:
            re, im = res.correlate(prm)
            time.sleep(0.001)
        check_close((re + 1j * im)[None], ref)
        assert res.info()["launches"] >= n0 + 2  # 0.5 s of calls against a 0.2 s lifetime (and the budget of 7)
    finally:
        res.close()


@pytest.mark.parametrize("bell", [2, 1])
@pytest.mark.parametrize("N,pollers", [(2048, 0), (16384, 0), (16384, 1), (65536, 0)])
def test_resident_survives_rings_that_race_with_its_exit(gat, N, pollers, bell):
    """Calls at random distances around the kernel's idle limit (and a small call budget): some rings arrive while the
    master is leaving -- served by some workgroups, by none, or by a kernel that is started for them.  Every call must
    return the same bits, none may hang (the host's own deadline would turn a lost ring into an error)."""
    g = gat
    rng = np.random.default_rng(N + pollers)
    case = make_case(21, N=N, M=4, L=3, K=1, B=1)
    ctx, res = _open(g, case, idle_us=300, life_ms=50, max_calls=37, host_pollers=pollers, doorbell=bell)
    try:
        prm = _params(g, case, 0)
        first = tuple(a.copy() for a in res.correlate(prm))
        check_close((first[0] + 1j * first[1])[None], oracle_result(case))
        for i in range(400):
            gap = rng.uniform(0.0, 600e-6)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < gap:
                pass
            re, im = res.correlate(prm)
            assert np.array_equal(re, first[0]) and np.array_equal(im, first[1]), i
        info = res.info()
        assert info["calls"] == 401 and info["launches"] > 5, info  # it did leave and come back many times
    finally:
        res.close()


def test_resident_is_parked_by_free_and_invalidated_by_new_codes(gat):
    g = gat
    import ctypes as C
    case = make_case(6, N=2048, M=1, L=3, K=1, B=1)
    ctx, res = _open(g, case)
    try:
        prm = _params(g, case, 0)
        res.correlate(prm)
        p = C.c_void_p()
        ctx.check(ctx.lib.gat_malloc(ctx._h, 4096, C.byref(p)), "gat_malloc")
        t0 = time.time()
        ctx.check(ctx.lib.gat_free(ctx._h, p), "gat_free")  # must not wait for the kernel's idle limit (5 ms default) for long
        assert time.time() - t0 < 0.5
        assert res.info()["running"] == 0
        re, im = res.correlate(prm)
        check_close((re + 1j * im)[None], oracle_result(case))
        # a new code table: the correlator answers GAT_ERR_STATE until it is opened again
        other = case["codes"].copy()
        other[0, 0] = -other[0, 0]
        ctx.set_codes(other)
        with pytest.raises(g._lib.GatError) as e:
            res.correlate(prm)
        assert e.value.status == 3
    finally:
        res.close()
    ctx.set_codes(case["codes"])


def test_resident_rejects_what_it_cannot_serve(gat):
    g = gat
    import torch
    ctx = g.get_context()
    case = make_case(8, N=4096, M=2, L=3, K=1, B=1)
    ctx.set_codes(case["codes"])
    dev = ctx.device
    re = torch.zeros((2, 4098), device=dev)
    torch.cuda.synchronize()
    L = g._lib
    cases = [
        (L.SignalDesc(re.data_ptr(), re.data_ptr(), g.GAT_LAYOUT_PLANAR, 2, 4096, 4096, 4096, 0), 17, [-1, 0, 1], 4),   # seventeen channels
        (L.SignalDesc(re.data_ptr(), re.data_ptr(), g.GAT_LAYOUT_PLANAR, 1, 4098, 4098, 4098, 0), 1, [-1, 0, 1], 4),    # ragged block length
        (L.SignalDesc(re.data_ptr(), re.data_ptr(), g.GAT_LAYOUT_PLANAR, 1, 4096, 4096, 4096, 0), 1, [-2000, 0, 2000], 4),  # taps of two launches (span > 2048)
    ]
    for desc, K, shifts, want in cases:
        with pytest.raises(L.GatError) as e:
            ctx.open_resident(desc, K, shifts, 4.096e6)
        assert e.value.status == want, (K, shifts, e.value)
    # bad calls on a good correlator
    desc = L.SignalDesc(re.data_ptr(), re.data_ptr(), g.GAT_LAYOUT_PLANAR, 1, 4096, 4096, 4096, 0)
    with ctx.open_resident(desc, 1, [-1, 0, 1], 4.096e6) as res:
        prm = _params(g, case, 0)
        for off in (-4, 3):
            with pytest.raises(L.GatError):
                res.correlate(prm, block_offset=off)
        bad = prm.copy()
        bad["prn"] = 99
        with pytest.raises(L.GatError) as e:
            res.correlate(bad)
        assert e.value.status == 2
        res.correlate(prm)  # still alive


def test_context_close_takes_its_resident_correlators_along(gat):
    g = gat
    import torch
    ctx = g.Context(0, "own")
    case = make_case(9, N=2048, M=4, L=3, K=1, B=1)
    ctx.set_codes(case["codes"])
    re = torch.from_numpy(case["re"]).to(ctx.device)
    im = torch.from_numpy(case["im"]).to(ctx.device)
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, 4, 2048, 2048, 2048, 0)
    res = ctx.open_resident(desc, 1, case["shifts"], case["fs"])
    r, i = res.correlate(_params(g, case, 0))
    check_close((r + 1j * i)[None], oracle_result(case))
    t0 = time.time()
    ctx.close()  # the kernel is asked to leave; nothing hangs
    assert time.time() - t0 < 1.0
    res.close()  # a no-op now


def test_resident_call_is_faster_than_launch_and_wait(gat):
    """What it is for: a call through the doorbell against the ordinary call + sync (completion-flag path, own stream) on
    the reference grid's small points.  Medians over 300 calls; asserted loosely (the point is the protocol, the numbers
    are in profiles/ and DESIGN.md)."""
    g = gat
    import torch
    ctx = g.get_context(own_stream=True)
    rows = []
    for N, M in ((2048, 4), (16384, 4)):
        case = make_case(3, N=N, M=M, L=3, K=1, B=1)
        ctx.set_codes(case["codes"])
        dev = ctx.device
        re = torch.from_numpy(case["re"]).to(dev)
        im = torch.from_numpy(case["im"]).to(dev)
        o_re = torch.zeros((1, 1, 3, M), device=dev)
        o_im = torch.zeros_like(o_re)
        torch.cuda.synchronize()
        desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
        prm = _params(g, case, 0)
        pdev = ctx.params_to_device(prm)
        torch.cuda.synchronize()
        call = ctx.prepared_call(desc, pdev, 1, 1, case["shifts"], case["fs"], o_re, o_im)
        t_ord = []
        for r in range(350):
            t0 = time.perf_counter()
            call()
            ctx.sync()
            t_ord.append(time.perf_counter() - t0)
        with ctx.open_resident(desc, 1, case["shifts"], case["fs"]) as res:
            t_res = []
            for r in range(350):
                t0 = time.perf_counter()
                res.correlate(prm)
                t_res.append(time.perf_counter() - t0)
            a, b = res.correlate(prm)
            check_close((a + 1j * b)[None], oracle_result(case))
        mo, mr = np.median(t_ord[50:]) * 1e6, np.median(t_res[50:]) * 1e6
        rows.append((N, M, mo, mr))
        print(f"single block N={N} M={M}: ordinary call + sync {mo:.1f} us, resident call {mr:.1f} us (Python host layer)")
    for N, M, mo, mr in rows:
        assert mr < mo, rows


# ---- round 5: housekeeping around device-wide waits, room on the device, buffer bounds, bound code tables ---------------------
def test_park_all_lets_a_device_wide_wait_return_at_once(gat):
    """A resident kernel with a long idle limit would make torch.cuda.synchronize() sit that limit out;
    ctx.device_synchronize() (gat_resident_park_all first) returns at once, and the next call starts the kernel again."""
    g = gat
    import torch
    case = make_case(21, N=4096, M=4, L=3, K=2, B=1)
    ctx, res = _open(g, case, idle_us=400000, life_ms=20000)
    with res:
        prm = _params(g, case, 0)
        r0 = np.array(res.correlate(prm)[0], copy=True)
        assert res.info()["running"] == 1
        t0 = time.perf_counter()
        ctx.device_synchronize()
        dt = time.perf_counter() - t0
        assert dt < 0.1, f"device-wide wait took {dt * 1e3:.1f} ms behind a parked resident kernel"
        info = res.info()
        assert info["running"] == 0 and info["last_exit"] == 1  # asked to leave
        r1 = res.correlate(prm)[0]  # starts it again
        assert np.array_equal(r0, r1) and res.info()["launches"] == 2


def test_open_refuses_what_the_device_cannot_hold_at_once(gat):
    """Every workgroup of every open resident correlator has to be ON the device for a call to complete: an open beyond that
    answers GAT_ERR_UNSUPPORTED at once (it used to loop until the call's 3 s deadline), and room comes back with a close."""
    g = gat
    import torch
    ctx = g.Context(0, "own")
    try:
        case = make_case(22, N=262144, M=4, L=3, K=1, B=1)
        ctx.set_codes(case["codes"])
        re = torch.from_numpy(case["re"]).to(ctx.device)
        im = torch.from_numpy(case["im"]).to(ctx.device)
        torch.cuda.synchronize()
        desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, 4, 262144, 262144, 262144, 0)
        cus = ctx.device_info()["num_cus"]
        opened, refused = [], None
        for _ in range(2 * cus // 100 + 8):  # each asks for up to 200 workgroups
            try:
                opened.append(ctx.open_resident(desc, 1, case["shifts"], case["fs"], max_workgroups=200, idle_us=200000))
            except g.GatError as e:
                refused = e
                break
        assert refused is not None and refused.status == 4, (len(opened), refused)
        assert 1 <= len(opened) and sum(r.info()["workgroups"] for r in opened) <= 8 * cus
        ref = oracle_result(case)
        r, i = opened[0].correlate(_params(g, case, 0))  # the ones that were admitted answer
        check_close((r + 1j * i)[None], ref)
        opened.pop().close()
        opened.append(ctx.open_resident(desc, 1, case["shifts"], case["fs"], max_workgroups=200, idle_us=200000))  # room again
        r, i = opened[-1].correlate(_params(g, case, 0))
        check_close((r + 1j * i)[None], ref)
    finally:
        ctx.close()


def test_python_layer_refuses_blocks_outside_the_buffer(gat):
    g = gat
    import torch
    case = make_case(23, N=2048, M=2, L=3, K=1, B=3)
    ctx = g.get_context()
    ctx.set_codes(case["codes"])
    re = torch.from_numpy(case["re"]).to(ctx.device)
    im = torch.from_numpy(case["im"]).to(ctx.device)
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, 2, 2048, 3 * 2048, 2048, 0)
    with ctx.open_resident(desc, 1, case["shifts"], case["fs"], buffer_samples=3 * 2048) as res:
        res.correlate(_params(g, case, 2), block_offset=2 * 2048)  # the last block: fine
        with pytest.raises(ValueError):
            res.correlate(_params(g, case, 2), block_offset=3 * 2048)  # one block past the allocation
    system = g.GPSL1(use_gpu=True)
    loop = g.ResidentTrackingLoop(system, np.array([1]), 2048, 2, case["fs"], case["shifts"], np.array([0.0]), np.array([0.0]), re=re, im=im)
    with loop:
        loop.run(3)
        with pytest.raises(ValueError):
            loop.run(2, start=2 * 2048)  # the second of them would end behind the buffer


def test_bound_code_table_cannot_be_edited_in_place(gat):
    """The context remembers the bound table by identity: an in-place edit must raise instead of silently keeping the old chips
    on the device; invalidate_codes() is the explicit way."""
    g = gat
    ctx = g.Context(0, "own")
    try:
        codes = np.array(g.GPSL1(use_gpu=True).codes, copy=True)
        ctx.set_codes(codes)
        assert codes.flags.writeable is False
        with pytest.raises(ValueError):
            codes[0, 0] = -codes[0, 0]
        codes.flags.writeable = True
        codes[0, :] = -codes[0, :]
        ctx.invalidate_codes()
        ctx.set_codes(codes)  # hashed again, uploaded
        assert ctx._codes_obj is codes
    finally:
        ctx.close()
