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
