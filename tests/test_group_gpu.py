"""Device groups (include/gat.h gat_group_*): satellite channels sharded over several contexts from one host thread,
signal replicated by peer copies, no collective.  On a one-GPU box the members share device 0 (own streams each);
the driver's 8-GPU node runs examples/gat_multi_gpu.c over its real devices unchanged.  Run with -m gpu."""
import subprocess

import numpy as np
import pytest

from tests.helpers import check_close, make_case, oracle_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gat():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def _single_device(g, case, lo, cnt):
    """Channels [lo, lo + cnt) of the case on one ordinary context -> complex [B, cnt, L, M]."""
    import torch
    sysobj = g.GNSSDICT[case["system"]](use_gpu=True)
    dev = g.get_context().device
    op = g.StreamCorrelator(sysobj, case["N"], case["M"], case["B"], cnt, case["shifts"], case["fs"])
    p = case["prm"][:, lo:lo + cnt]
    op.set_params(g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"],
                                p["carrier_phase_cycles"]))
    op(torch.from_numpy(case["re"]).to(dev), torch.from_numpy(case["im"]).to(dev))
    return op.result()


@pytest.mark.parametrize("members,K,M,N", [(2, 5, 4, 4000), (3, 8, 16, 6000), (2, 1, 2, 2500), (4, 4, 1, 2048)])
def test_group_matches_single_device_and_oracle(gat, members, K, M, N):
    """Ragged channel split (5 over 2, 8 over 3, 1 over 2: one member idles), signal uploaded to member 0 only and
    replicated to the peers: the gathered result is bit-identical to the same shards run on an ordinary context and
    within 1e-5 of the FP64 oracle."""
    import torch
    g = gat
    case = make_case(100 + K + M, N=N, M=M, K=K, B=3, L=3)
    B, L = case["B"], case["L"]
    grp = g.DeviceGroup([0] * members)
    assert grp.size == members
    grp.set_codes(case["codes"])
    dev = torch.device("cuda", 0)
    # every member owns its copy of the signal; only member 0's is filled by the host
    res, ims = [], []
    for r in range(members):
        res.append(torch.from_numpy(case["re"]).to(dev) if r == 0 else torch.full((M, B * N), float("nan"), device=dev))
        ims.append(torch.from_numpy(case["im"]).to(dev) if r == 0 else torch.full((M, B * N), float("nan"), device=dev))
    torch.cuda.synchronize()  # the tensors were filled on torch's stream; the members run on their own streams
    grp.replicate(0, res)
    grp.replicate(0, ims)
    descs, o_re, o_im = [], [], []
    bounds = [grp.shard(K, r) for r in range(members)]
    assert sum(c for _, c in bounds) == K and [b[0] for b in bounds] == [g.ShardPlan(K, members, r).lo for r in range(members)]
    for r in range(members):
        d = g._lib.SignalDesc(res[r].data_ptr(), ims[r].data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, B * N, N, 0)
        descs.append(d)
        o_re.append(torch.full((B, max(bounds[r][1], 1), L, M), float("nan"), device=dev))
        o_im.append(torch.full((B, max(bounds[r][1], 1), L, M), float("nan"), device=dev))
    torch.cuda.synchronize()
    prm = g.make_params(case["prm"]["prn0"], case["prm"]["code_freq_hz"], case["prm"]["carrier_freq_hz"],
                        case["prm"]["code_phase_chips"], case["prm"]["carrier_phase_cycles"])
    grp.correlate(descs, prm, case["shifts"], case["fs"], o_re, o_im)
    got = grp.gather(o_re, o_im, B, K, L, M)
    assert np.isfinite(got.view(np.float32)).all()
    check_close(got, oracle_result(case), what="group vs oracle")
    for r, (lo, cnt) in enumerate(bounds):
        if cnt:
            one = _single_device(g, case, lo, cnt)
            assert np.array_equal(got[:, lo:lo + cnt].view(np.float32), one.astype(np.complex64).view(np.float32)), f"member {r}"
    for t in (res[1], ims[1]):  # the peer copy really happened
        assert torch.equal(t, res[0] if t is res[1] else ims[0])
    grp.close()


def test_memcpy_peer_orders_after_the_source_stream(gat):
    """gat_memcpy_peer copies after everything enqueued on the SOURCE member's stream: a signal generated there by
    gat_gen_signal (asynchronous) arrives complete at the peer without any host synchronisation in between."""
    import ctypes as C

    import torch
    g = gat
    lib = g.load_library()
    grp = g.DeviceGroup([0, 0])
    sysobj = g.GPSL1(use_gpu=True)
    grp.set_codes(sysobj.codes)
    ctxs = []
    for r in range(2):
        h = C.c_void_p()
        assert lib.gat_group_ctx(grp._h, r, C.byref(h)) == 0
        ctxs.append(h)
    N, M, B = 200000, 4, 8
    dev = torch.device("cuda", 0)
    src = [torch.zeros((M, B * N), device=dev) for _ in range(2)]
    dst = [torch.full((M, B * N), float("nan"), device=dev) for _ in range(2)]
    prm = g.make_params(np.zeros((B, 1), dtype=np.int32), 1.023e6, 1500.0, 0.0, 0.0)
    prm_dev = torch.from_numpy(np.ascontiguousarray(prm).view(np.uint8).reshape(-1).copy()).to(dev)
    torch.cuda.synchronize()
    rc = lib.gat_gen_signal(ctxs[0], src[0].data_ptr(), src[1].data_ptr(), 0, N, M, B * N, N, B, 1, prm_dev.data_ptr(),
                            float(N / 1e-3), 1.0)
    assert rc == 0
    for i in range(2):
        assert lib.gat_memcpy_peer(ctxs[1], dst[i].data_ptr(), ctxs[0], src[i].data_ptr(), src[i].numel() * 4) == 0
    grp.sync()
    assert torch.equal(dst[0], src[0]) and torch.equal(dst[1], src[1])
    assert float(src[0].abs().max()) > 0.5
    grp.close()


def test_source_may_be_refilled_right_after_replication(gat):
    """A streaming receiver refills its ingest buffer for the next block as soon as gat_group_replicate has returned:
    the refill (enqueued on the SOURCE member's stream) must wait for the asynchronous peer copies that still read the
    buffer -- gat_memcpy_peer orders both ways.  Block i is generated with carrier frequency f_i, replicated, and the
    buffer is overwritten at once (memset + the next block's generator); every replica must hold block i, not a torn mix."""
    import ctypes as C

    import torch
    g = gat
    lib = g.load_library()
    grp = g.DeviceGroup([0, 0, 0])
    sysobj = g.GPSL1(use_gpu=True)
    grp.set_codes(sysobj.codes)
    ctxs = []
    for r in range(3):
        h = C.c_void_p()
        assert lib.gat_group_ctx(grp._h, r, C.byref(h)) == 0
        ctxs.append(h)
    N, M, rounds = 1 << 21, 4, 6  # 32 MB per plane: the copies take far longer than the calls that enqueue the refill
    dev = torch.device("cuda", 0)
    src = [torch.zeros((M, N), device=dev) for _ in range(2)]
    keep = [[[torch.empty((M, N), device=dev) for _ in range(2)] for _ in range(rounds)] for _ in range(2)]  # [peer][round][plane]
    prms = []
    for i in range(rounds):
        prm = g.make_params(np.zeros((1, 1), dtype=np.int32), 1.023e6, 1000.0 * (i + 1), 0.0, 0.0)
        prms.append(torch.from_numpy(np.ascontiguousarray(prm).view(np.uint8).reshape(-1).copy()).to(dev))
    fs = float(N / 1e-3)
    torch.cuda.synchronize()
    for i in range(rounds):
        assert lib.gat_gen_signal(ctxs[0], src[0].data_ptr(), src[1].data_ptr(), 0, N, M, N, N, 1, 1, prms[i].data_ptr(), fs, 1.0) == 0
        for peer in (1, 2):
            for pl in range(2):
                assert lib.gat_memcpy_peer(ctxs[peer], keep[peer - 1][i][pl].data_ptr(), ctxs[0], src[pl].data_ptr(), src[pl].numel() * 4) == 0
        # no synchronisation: the source stream overwrites the buffer at once
        assert lib.gat_memset(ctxs[0], src[0].data_ptr(), 0xFF, src[0].numel() * 4) == 0
        assert lib.gat_memset(ctxs[0], src[1].data_ptr(), 0xFF, src[1].numel() * 4) == 0
    grp.sync()
    # what block i looked like: generate it again, undisturbed
    for i in range(rounds):
        assert lib.gat_gen_signal(ctxs[0], src[0].data_ptr(), src[1].data_ptr(), 0, N, M, N, N, 1, 1, prms[i].data_ptr(), fs, 1.0) == 0
        grp.sync()
        for peer in range(2):
            for pl in range(2):
                assert torch.equal(keep[peer][i][pl], src[pl]), f"round {i} peer {peer + 1} plane {pl}: torn replica"
    grp.close()


@pytest.mark.parametrize("args", [["1", "4", "4"], ["2", "4", "4"], ["3", "2", "2"]])
def test_multi_gpu_c_example(gat, args):
    """examples/gat_multi_gpu.c from plain C: members over the visible devices (more members than devices wrap around),
    ingest on member 0, peer replication, sharded launch, gather; exits 0 only when the gathered result is bit-identical
    to the shard-by-shard single-device result and every PRN found its signal."""
    from gpuacceleratedtracking_amd import build
    exe = build.build_c_example(name="gat_multi_gpu")
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "BIT-IDENTICAL" in r.stdout


def test_group_argument_errors(gat):
    import ctypes as C
    g = gat
    lib = g.load_library()
    h = C.c_void_p()
    assert lib.gat_group_create(0, None, C.byref(h)) == 1          # GAT_ERR_ARG
    bad = (C.c_int32 * 1)(99)
    assert lib.gat_group_create(1, bad, C.byref(h)) == 2 and not h.value  # GAT_ERR_RANGE from gat_create
    n = C.c_int32()
    assert lib.gat_device_count(C.byref(n)) == 0 and n.value >= 1
    grp = g.DeviceGroup([0])
    lo, cnt = grp.shard(7, 0)
    assert (lo, cnt) == (0, 7)
    with pytest.raises(g.GatError):
        grp.shard(7, 1)
    grp.close()


def test_native_latency_example_and_completion_flag(gat):
    """examples/gat_latency.c: the harness's timed body from plain C on a library-owned stream -- every grid point runs the
    host-parameter call, the device-parameter call (completion flag: gat_sync spins on the sequence number the launch's
    last workgroup stores into pinned host memory) and the hipGraph replay; each point prints the prompt correlation, which
    must be N (noise-free signal, zero phases): a flag that fired before the results were out would show here.  The same
    call through a resident correlator (no launch: rung into a kernel that stays on the device) prints its prompt too."""
    import re
    from gpuacceleratedtracking_amd import build
    exe = build.build_c_example(name="gat_latency")
    r = subprocess.run([exe, "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("#")]
    assert len(rows) == 8 * 2 * 2 + 4 * 2 * 1, r.stdout  # the reference's GPS L1 grid and its GPS L5 grid
    assert sum(ln.split()[0] == "GPSL5" for ln in rows) == 8
    for ln in rows:
        n = int(ln.split()[1])
        m = re.search(r"\(prompt (\d+) / (\d+)\)", ln)
        assert m and int(m.group(1)) == n, ln
        dev_min = float(ln.split("|")[2].split("/")[0])
        assert 3.0 < dev_min < 200.0, ln  # microseconds: a call + sync is neither free nor a stall
        res_min, wgs = float(ln.split("|")[6].split("/")[0]), int(re.search(r"\(\s*(\d+)\)", ln.split("|")[6]).group(1))
        assert wgs > 0 and int(m.group(2)) == n and 1.0 < res_min < 200.0, ln


def test_sync_after_flagged_launch_sees_results_and_later_work_falls_back(gat):
    """gat_sync on a library-owned stream: after a flagged correlate the results are visible to a device-to-host copy; work
    enqueued after the flagged launch (a memset of the outputs) makes the next gat_sync wait for the stream instead of the flag."""
    import ctypes as C
    g = gat
    lib = g.load_library()
    h = C.c_void_p()
    assert lib.gat_create(0, C.c_void_p(-1), C.byref(h)) == 0  # GAT_OWN_STREAM
    system = g.GPSL1(use_gpu=True)
    codes = np.ascontiguousarray(system.codes, dtype=np.int8)
    assert lib.gat_set_codes(h, codes.ctypes.data_as(C.POINTER(C.c_int8)), codes.shape[1], codes.shape[0]) == 0
    N, M, L = 2500, 4, 3
    fs = N / 1e-3
    bufs = {}
    for name, nbytes in (("re", 4 * N * M), ("im", 4 * N * M), ("prm", 40), ("o_re", 4 * M * L), ("o_im", 4 * M * L)):
        p = C.c_void_p()
        assert lib.gat_malloc(h, nbytes, C.byref(p)) == 0
        bufs[name] = p
    prm = g.make_params(np.zeros((1, 1), dtype=np.int32), 1.023e6, 1500.0, 0.0, 0.0)
    prm = np.ascontiguousarray(prm)
    assert lib.gat_memcpy_h2d(h, bufs["prm"], prm.ctypes.data_as(C.c_void_p), 40) == 0
    assert lib.gat_gen_signal(h, bufs["re"], bufs["im"], 0, N, M, N, N, 1, 1, bufs["prm"], fs, 1.0) == 0
    desc = g._lib.SignalDesc(bufs["re"].value, bufs["im"].value, 0, M, N, N, N, 0)
    shifts = (C.c_int32 * L)(-1, 0, 1)
    out = np.zeros(M * L, dtype=np.float32)
    for rep in range(50):
        assert lib.gat_memset(h, bufs["o_re"], 0xFF, 4 * M * L) == 0     # NaN pattern
        assert lib.gat_downconvert_and_correlate_dev(h, C.byref(desc), bufs["prm"], 1, 1, L, shifts, fs, bufs["o_re"], bufs["o_im"], 0) == 0
        assert lib.gat_sync(h) == 0                                        # flag path
        assert lib.gat_memcpy_d2h(h, out.ctypes.data_as(C.c_void_p), bufs["o_re"], 4 * M * L) == 0
        assert np.allclose(out.reshape(L, M)[:, 0], [1476, 2500, 1476], rtol=1e-5), (rep, out)
    # newer work behind a flagged launch: gat_sync must not return on the (already reached) flag
    assert lib.gat_downconvert_and_correlate_dev(h, C.byref(desc), bufs["prm"], 1, 1, L, shifts, fs, bufs["o_re"], bufs["o_im"], 0) == 0
    assert lib.gat_memset(h, bufs["o_re"], 0, 4 * M * L) == 0
    assert lib.gat_sync(h) == 0
    assert lib.gat_memcpy_d2h(h, out.ctypes.data_as(C.c_void_p), bufs["o_re"], 4 * M * L) == 0
    assert (out == 0).all()
    for p in bufs.values():
        lib.gat_free(h, p)
    assert lib.gat_destroy(h) == 0


@pytest.mark.parametrize("N,M", [(2500, 4), (65536, 4), (20000, 1), (2046, 1), (65538, 1)])
def test_completion_flag_results_are_out_when_the_flag_is(gat, N, M):
    """The completion flag must not reach the host before the results are in memory.  The copies in the test above are
    stream-ordered behind the kernel and would pass either way; here the outputs are read by PyTorch's stream, which is NOT
    ordered with the context's own stream -- the read is correct only if gat_sync returned after the results were written
    back.  The carrier phase changes from call to call (a rotation of every accumulator), so a stale or half-written result
    shows as the previous call's rotation; one-workgroup launches (N = 2500: no arrival counter), a split block with the
    second stage carrying the flag (N = 65536), ragged lengths whose tail launch carries it (N = 2046, 65538), and
    host-parameter records inside the kernel arguments."""
    import torch
    from gpuacceleratedtracking_amd import _lib
    from gpuacceleratedtracking_amd.context import Context
    g = gat
    ctx = Context(0, "own")
    dev = ctx.device
    system = g.GPSL1(use_gpu=True)
    ctx.set_codes(system.codes)
    fs, L = N / 1e-3, 3
    sig = g.gen_signal(system, 1, 1500.0, N, num_ants=g.NumAnts(M))[0]
    re, im = sig.re.reshape(M, N).contiguous(), sig.im.reshape(M, N).contiguous()
    desc = _lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
    o_re = torch.zeros(L * M, dtype=torch.float32, device=dev)
    o_im = torch.zeros_like(o_re)
    shifts = [-1, 0, 1]
    torch.cuda.synchronize()

    def call(phase):
        prm = g.make_params(np.zeros((1, 1), dtype=np.int32), 1.023e6, 1500.0, 0.0, phase)
        ctx.downconvert_and_correlate(desc, prm, 1, 1, shifts, fs, o_re, o_im)
        ctx.sync()  # completion flag
        return (o_re.cpu().numpy() + 1j * o_im.cpu().numpy()).copy()  # PyTorch's stream: not ordered with ctx's

    base = call(0.0)
    assert abs(base.reshape(L, M)[1, 0] - N) < 1e-3 * N
    # (N = 2046 / 65538: ragged block lengths -- the flag is carried by dc_tail_kernel, the call's last launch; a single
    # antenna, so that the unpadded antenna stride does not matter)
    assert ctx.last_launch_info()["finalize_launched"] == (0 if N in (2500, 2046) else 1)
    assert ctx.last_launch_info()["vec"] == 4
    bad = 0
    for i in range(1, 1500):
        phase = (i * 0.0371) % 1.0
        got = call(phase)
        want = base * np.exp(-2j * np.pi * phase)
        if not np.allclose(got, want, rtol=0, atol=2e-4 * N):
            bad += 1
    assert bad == 0, f"{bad} of 1499 calls returned before their results were visible"
    ctx.close()
