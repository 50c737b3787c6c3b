"""GPU tests of the closed tracking-loop step.  Parity: the device kernel
against the oracle's numpy restatement of the same equations (unpinned against the reference:
Tracking.jl is not available).  Behaviour: the loop pulls in a Doppler and a code-phase offset on a
synthetic signal and then tracks it."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    g.load_library()
    return g


def _cfg_dict(cfg):
    return {n: getattr(cfg, n) for n, _ in cfg._fields_}


def test_update_matches_oracle_restatement(g):
    import torch
    system = g.GPSL1()
    N, M, K, fs = 4000, 2, 5, 4e6
    shifts = np.array([-2, 0, 2], dtype=np.int32)
    rng = np.random.default_rng(9)
    loop = g.TrackingLoop(system, [1, 5, 9, 13, 17], N, M, fs, shifts, init_carrier_doppler=rng.uniform(-3e3, 3e3, K),
                          init_code_phase=rng.uniform(0, 1023, K), if_hz=1.0e5,
                          init_carrier_phase=rng.uniform(0, 1, K))
    cfg = _cfg_dict(loop.config)
    state = {n: loop.state()[n].copy() for n in loop.state().dtype.names}
    cur = loop.params()[...].reshape(-1).copy()
    ocur = oracle.make_params(cur["prn"], cur["code_freq_hz"], cur["carrier_freq_hz"], cur["code_phase_chips"],
                              cur["carrier_phase_cycles"])
    dev = loop.ctx.device
    for it in range(4):  # several steps so that the filter integrators are exercised
        acc = (rng.standard_normal((K, 3, M)) + 1j * rng.standard_normal((K, 3, M))).astype(np.complex64) * 1000
        loop.out_re.copy_(torch.from_numpy(acc.real.copy()).to(dev).reshape(loop.out_re.shape))
        loop.out_im.copy_(torch.from_numpy(acc.imag.copy()).to(dev).reshape(loop.out_im.shape))
        import ctypes as C
        c, n = loop._params[loop._cur], loop._params[1 - loop._cur]
        rc = loop.ctx.lib.gat_tracking_update(loop.ctx._h, C.c_void_p(loop.out_re.data_ptr()),
                                              C.c_void_p(loop.out_im.data_ptr()), K, M, C.byref(loop.config),
                                              C.c_void_p(loop._state.data_ptr()), C.c_void_p(c.data_ptr()),
                                              C.c_void_p(n.data_ptr()))
        assert rc == 0
        loop._cur = 1 - loop._cur
        ocur, state = oracle.np_tracking_update(acc, cfg, state, ocur)
        got = loop.params().reshape(-1)
        gst = loop.state()
        for f_o, f_g in (("code_freq_hz", "code_freq_hz"), ("carrier_freq_hz", "carrier_freq_hz"),
                         ("code_phase_chips", "code_phase_chips"), ("carrier_phase_cycles", "carrier_phase_cycles")):
            assert np.allclose(got[f_g], ocur[f_o], rtol=1e-12, atol=1e-9), (it, f_o)
        for name in state:
            assert np.allclose(gst[name], state[name], rtol=1e-10, atol=1e-9), (it, name)


def _run_closed_loop(g, prns, true_dop, true_tau0, true_phi0, nblk=1500):
    system = g.GPSL1()
    N, M, fs, fc = 4000, 2, 4e6, 1.023e6
    K = prns.size
    fcode = fc * (1 + true_dop / 1575.42e6)
    # truth signal: every block continues code and carrier phase exactly
    b = np.arange(nblk, dtype=np.float64)[:, None]
    tau = np.mod(true_tau0[None, :] + fcode[None, :] * (N / fs) * b, 1023.0)
    phi = np.mod(true_phi0[None, :] + true_dop[None, :] * (N / fs) * b, 1.0)
    prm_sig = g.make_params(prns - 1, fcode, true_dop, tau, 2 * np.pi * phi, shape=(nblk, K))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    loop = g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=true_dop + 12.0,
                          init_code_phase=true_tau0 + 0.12, init_carrier_phase=0.0, dll_bandwidth_hz=4.0)
    for i in range(nblk):
        loop.step(re, im, start=i * N)
    st, p = loop.state(), loop.params().reshape(-1)
    tau_true_end = np.mod(true_tau0 + fcode * (N / fs) * nblk, 1023.0)
    dtau = np.abs(((p["code_phase_chips"] - tau_true_end + 511.5) % 1023.0) - 511.5)
    return st, dtau, loop.accumulators(), N


def test_closed_loop_single_satellite_converges(g):
    """One satellite, 2 antennas, fs = 4 MHz, 1 ms blocks: start 12 Hz and 0.12 chip off; after
    1.5 s the 3rd-order PLL (18 Hz) / 2nd-order DLL (4 Hz) sit on the truth."""
    st, dtau, acc, N = _run_closed_loop(g, np.array([7]), np.array([-2210.0]), np.array([511.9]), np.array([0.6]))
    assert abs(st["carrier_doppler_hz"][0] + 2210.0) < 0.2
    assert dtau[0] < 0.02
    assert abs(st["last_pll_error_cycles"][0]) < 5e-3 and abs(st["last_dll_error_chips"][0]) < 0.02
    assert (np.abs(acc[0, 1, :]) > 0.97 * N).all()  # prompt ~ N on every antenna
    assert (np.abs(acc[0, 1, :].imag) < 0.04 * N).all()  # all of it in phase (Costas lock: +-I)


def test_closed_loop_four_satellites_track(g):
    """Four equal-power satellites: each channel sees the other three as cross-correlation noise
    (~ -24 dB), so the instantaneous errors jitter but the loops stay locked on the truth."""
    true_dop = np.array([1234.5, -2210.0, 310.0, -95.0])
    st, dtau, acc, N = _run_closed_loop(g, np.array([2, 7, 19, 30]), true_dop, np.array([100.3, 511.9, 900.05, 17.6]),
                                        np.array([0.1, 0.6, 0.35, 0.9]))
    assert np.abs(st["carrier_doppler_hz"] - true_dop).max() < 3.0
    assert dtau.max() < 0.06
    assert np.abs(st["last_pll_error_cycles"]).max() < 0.08 and np.abs(st["last_dll_error_chips"]).max() < 0.15
    assert (np.abs(acc[:, 1, :]) > 0.85 * N).all()


def test_native_run_equals_stepwise_loop(g):
    """gat_tracking_run (all blocks from one native call) enqueues exactly the launches of step() x blocks: parameters,
    loop state and every block's accumulators are bit-identical."""
    system = g.GPSL1()
    N, M, fs, nblk = 4000, 2, 4e6, 64
    prns = np.array([3, 11, 26])
    dop = np.array([850.0, -1400.0, 40.0])
    prm_sig = g.make_params(prns - 1, 1.023e6, dop, [[10.0, 400.5, 900.25]], 0.0, shape=(nblk, 3))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)

    def make():
        return g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop + 5.0,
                              init_code_phase=np.array([10.1, 400.4, 900.3]), dll_bandwidth_hz=4.0)
    a, b = make(), make()
    hist = []
    for i in range(nblk):
        a.step(re, im, start=i * N)
        hist.append(a.accumulators())
    acc_re, acc_im = b.run(re, im, nblk)
    got = (acc_re.cpu().numpy() + 1j * acc_im.cpu().numpy()).astype(np.complex64)
    assert np.array_equal(np.stack(hist).view(np.float32), got.view(np.float32))
    assert a.params().tobytes() == b.params().tobytes() and a.state().tobytes() == b.state().tobytes()
    assert b.blocks_done == nblk
    c = make()  # keep=False: only the last block's accumulators, same parameters
    c.run(re, im, nblk, keep=False)
    assert c.params().tobytes() == a.params().tobytes()
    assert np.array_equal(c.accumulators().view(np.float32), hist[-1].view(np.float32))


def test_native_run_graph_replay(g):
    """GAT_FLAG_GRAPH: the second and later calls with the same buffers replay the recorded hipGraph -- the results
    equal the eager run's (the parameter buffers swap an even number of times per call, so every call starts from
    buffer A; state and parameters evolve on the device across replays exactly as across eager calls)."""
    import torch
    system = g.GPSL1()
    N, M, fs, nblk = 4000, 2, 4e6, 32  # even: the ping-pong returns to buffer A after every call
    prns = np.array([5, 17])
    dop = np.array([-600.0, 1900.0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        prm_sig = g.make_params(prns - 1, 1.023e6, dop, [[77.0, 640.5]], 0.0, shape=(nblk, 2))
        re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
        shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)

        def make():
            return g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop + 3.0,
                                  init_code_phase=np.array([77.1, 640.4]), dll_bandwidth_hz=4.0)
        eager, graph = make(), make()
        assert graph.ctx.stream.cuda_stream != 0
        out = (torch.empty((nblk, 2, 3, M), device=graph.ctx.device), torch.empty((nblk, 2, 3, M), device=graph.ctx.device))
        for rep in range(3):  # same signal three times over: call 1 records, calls 2 and 3 replay
            e_re, e_im = eager.run(re, im, nblk)
            graph.run(re, im, nblk, graph=True, out=out)
            graph.ctx.sync()
            assert torch.equal(e_re, out[0]) and torch.equal(e_im, out[1]), rep
            assert eager.params().tobytes() == graph.params().tobytes() and eager.state().tobytes() == graph.state().tobytes()


def test_graph_replay_survives_scratch_growth_and_code_rebinding(g):
    """a recorded hipGraph bakes in the library's split-partials buffer and code tables.  After
    graph runs, (1) a larger correlate on the same context grows (re-allocates) that buffer, (2) another system's table
    is bound and the first one re-bound: the next graph-flagged runs must re-record instead of replaying launches that
    point at freed memory / stale tables -- and stay bit-identical to the eager loop."""
    import torch
    system = g.GPSL1()
    N, M, fs, nblk = 4000, 2, 4e6, 16
    prns = np.array([5, 17])
    dop = np.array([-600.0, 1900.0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        prm_sig = g.make_params(prns - 1, 1.023e6, dop, [[77.0, 640.5]], 0.0, shape=(nblk, 2))
        re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
        shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)

        def make():
            return g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop + 3.0,
                                  init_code_phase=np.array([77.1, 640.4]), dll_bandwidth_hz=4.0)
        eager, graph = make(), make()
        ctx = graph.ctx
        assert ctx.stream.cuda_stream != 0
        out = (torch.empty((nblk, 2, 3, M), device=ctx.device), torch.empty((nblk, 2, 3, M), device=ctx.device))

        def both(tag):
            e_re, e_im = eager.run(re, im, nblk)
            graph.run(re, im, nblk, graph=True, out=out)
            ctx.sync()
            assert torch.equal(e_re, out[0]) and torch.equal(e_im, out[1]), tag
            assert eager.params().tobytes() == graph.params().tobytes(), tag
        both("record")
        both("replay")
        # (1) grow the library's scratch buffer on the same context: one long block split over many workgroups
        big_n = 2_000_000
        big = g.StreamCorrelator(system, big_n, 4, 1, 2, shifts, fs, ctx=ctx)
        big.set_params(g.make_params(np.arange(2), 1.023e6, 1500.0, 0.0, 0.0, shape=(1, 2)))
        big(torch.zeros((4, big_n), device=ctx.device), torch.zeros((4, big_n), device=ctx.device))
        assert ctx.last_launch_info()["splits"] > 1
        ctx.sync()
        both("after scratch growth")
        both("replay after scratch growth")
        # (2) another system's tables on the same context, then back (TrackingLoop re-binds its table per run)
        l5 = g.StreamCorrelator(g.GPSL5(), 4096, 1, 1, 1, np.array([0], dtype=np.int32), 50e6, ctx=ctx)
        l5.set_params(g.make_params(0, 10.23e6, 0.0, 0.0, 0.0, shape=(1, 1)))
        l5(torch.ones((1, 4096), device=ctx.device), torch.zeros((1, 4096), device=ctx.device))
        ctx.sync()
        both("after another system's table")
        both("replay after another system's table")


def test_graph_in_flight_survives_knob_changes_and_scratch_growth_without_sync(g):
    """gat_tracking_run(GAT_FLAG_GRAPH) is asynchronous; a knob change (set_vector_tiling /
    set_matrix_core) or a scratch growth right behind it destroys the recorded graphs -- the library must drain the
    stream first.  No ctx.sync() between the graph run and what invalidates it; results stay bit-identical to eager."""
    import torch
    system = g.GPSL1()
    N, M, fs, nblk = 4000, 2, 4e6, 64
    prns = np.array([5, 17])
    dop = np.array([-600.0, 1900.0])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        prm_sig = g.make_params(prns - 1, 1.023e6, dop, [[77.0, 640.5]], 0.0, shape=(nblk, 2))
        re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
        shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)

        def make():
            return g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop + 3.0,
                                  init_code_phase=np.array([77.1, 640.4]), dll_bandwidth_hz=4.0)
        eager, graph = make(), make()
        ctx = graph.ctx
        out = (torch.empty((nblk, 2, 3, M), device=ctx.device), torch.empty((nblk, 2, 3, M), device=ctx.device))
        big_n = 2_000_000
        big = g.StreamCorrelator(system, big_n, 4, 1, 2, shifts, fs, ctx=ctx)
        big.set_params(g.make_params(np.arange(2), 1.023e6, 1500.0, 0.0, 0.0, shape=(1, 2)))
        zeros = torch.zeros((4, big_n), device=ctx.device)
        ctx.sync()
        invalidators = [lambda: ctx.set_vector_tiling(4, 4, 16), lambda: ctx.set_matrix_core(g.GAT_MC_AUTO),
                        lambda: big(zeros, zeros)]  # the last one grows the split-partials buffer (first time only)
        for i, inval in enumerate(invalidators * 2):
            graph.run(re, im, nblk, graph=True, out=out)   # record (or re-record)
            graph.run(re, im, nblk, graph=True, out=out)   # replay: 2 * nblk launches in flight ...
            inval()                                        # ... when the recorded graphs are dropped
            e_re, e_im = eager.run(re, im, nblk)
            e_re, e_im = eager.run(re, im, nblk)
            ctx.sync()
            assert torch.equal(e_re, out[0]) and torch.equal(e_im, out[1]), i
            assert eager.params().tobytes() == graph.params().tobytes(), i


def test_two_systems_share_one_context(g):
    """the code table is per-context state; operators of two systems created on the same context
    must each correlate against THEIR table, whatever the creation order (the reference passes `system` per call)."""
    import torch
    ctx = g.get_context()
    dev = ctx.device
    l1, l5 = g.GPSL1(use_gpu=True), g.GPSL5(use_gpu=True)
    N = 10230
    ops = {}
    for name, system, fc, fs in (("L1", l1, 1.023e6, 10.23e6), ("L5", l5, 10.23e6, 10.23e6)):
        op = g.StreamCorrelator(system, N, 1, 1, 1, np.array([0], dtype=np.int32), fs, ctx=ctx)
        op.set_params(g.make_params(4, fc, 0.0, 0.0, 0.0, shape=(1, 1)))
        ops[name] = (op, system, fc, fs)
    # signals: each system's own PRN 5 code at one sample per chip (L5) / ten samples per chip (L1) -> sum = N
    sig = {}
    for name, (op, system, fc, fs) in ops.items():
        idx = np.floor(np.arange(N) * fc / fs).astype(int) % system.codes.shape[1]
        sig[name] = torch.from_numpy(system.codes[4, idx].astype(np.float32))[None, :].to(dev)
    zero = torch.zeros((1, N), device=dev)
    for order in (("L1", "L5", "L1", "L5"), ("L5", "L5", "L1", "L1")):
        for name in order:
            op = ops[name][0]
            op(sig[name], zero)
            assert op.result()[0, 0, 0, 0].real == N, (order, name)


def test_host_closed_loop_through_the_resident_correlator(g):
    """The loop with the host in it (ResidentTrackingLoop: a call rung into a resident correlator + gat_tracking_update_host per
    block, as the reference's receiver closes its loops on the CPU) against the loop that never leaves the device
    (TrackingLoop.run), same signal and start values: both sit on the truth, and on each other -- the correlator outputs of
    the two differ in summation order only (1e-7), which the discriminators see as 1e-7 of a cycle."""
    system = g.GPSL1()
    N, M, fs, fc, nblk = 4000, 2, 4e6, 1.023e6, 600
    prns = np.array([3, 11, 25])
    true_dop = np.array([1500.0, -800.0, 2400.0])
    tau0, phi0 = np.array([200.4, 700.1, 33.3]), np.array([0.2, 0.7, 0.4])
    fcode = fc * (1 + true_dop / 1575.42e6)
    b = np.arange(nblk, dtype=np.float64)[:, None]
    prm_sig = g.make_params(prns - 1, fcode, true_dop, np.mod(tau0[None, :] + fcode[None, :] * (N / fs) * b, 1023.0),
                            2 * np.pi * np.mod(phi0[None, :] + true_dop[None, :] * (N / fs) * b, 1.0), shape=(nblk, prns.size))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    kw = dict(init_carrier_doppler=true_dop + 8.0, init_code_phase=tau0 + 0.1, init_carrier_phase=0.0, dll_bandwidth_hz=4.0)
    dev = g.TrackingLoop(system, prns, N, M, fs, shifts, **kw)
    dev.run(re, im, nblk, keep=False)
    dev.ctx.sync()
    import torch
    torch.cuda.synchronize()
    with g.ResidentTrackingLoop(system, prns, N, M, fs, shifts, re=re, im=im, idle_us=200000, **kw) as host:
        host.run(nblk)
        info = host.resident.info()
        hs, hp, hacc = host.state(), host.params(), host.accumulators()
    ds, dp = dev.state(), dev.params().reshape(-1)
    assert info["calls"] == nblk
    assert np.abs(hs["carrier_doppler_hz"] - true_dop).max() < 1.0 and np.abs(ds["carrier_doppler_hz"] - true_dop).max() < 1.0
    assert np.abs(hs["carrier_doppler_hz"] - ds["carrier_doppler_hz"]).max() < 1e-3
    assert np.abs(hp["code_phase_chips"] - dp["code_phase_chips"]).max() < 1e-4
    assert np.abs(hp["carrier_freq_hz"] - dp["carrier_freq_hz"]).max() < 1e-3
    assert (np.abs(hacc[:, 1, :]) > 0.85 * N).all()  # prompt ~ N on every antenna of every channel


def test_native_host_closed_loop_equals_block_by_block_stepping(g):
    """gat_resident_tracking_run (the {resident call, host update} loop from native code) against the same loop stepped block by
    block from Python: same calls in the same order -- parameters, loop state and every block's accumulators bit-identical;
    argument errors named."""
    import ctypes as C
    system = g.GPSL1()
    N, M, fs, fc, nblk = 4000, 4, 4e6, 1.023e6, 120
    prns = np.array([5, 17])
    true_dop = np.array([900.0, -2100.0])
    tau0, phi0 = np.array([11.7, 512.3]), np.array([0.1, 0.6])
    fcode = fc * (1 + true_dop / 1575.42e6)
    b = np.arange(nblk, dtype=np.float64)[:, None]
    prm_sig = g.make_params(prns - 1, fcode, true_dop, np.mod(tau0[None, :] + fcode[None, :] * (N / fs) * b, 1023.0),
                            2 * np.pi * np.mod(phi0[None, :] + true_dop[None, :] * (N / fs) * b, 1.0), shape=(nblk, prns.size))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    kw = dict(init_carrier_doppler=true_dop + 5.0, init_code_phase=tau0 + 0.05, init_carrier_phase=0.0, dll_bandwidth_hz=4.0)
    import torch
    torch.cuda.synchronize()
    with g.ResidentTrackingLoop(system, prns, N, M, fs, shifts, re=re, im=im, idle_us=200000, **kw) as a:
        accs = []
        for blk in range(nblk):
            a.step(blk * N)
            accs.append(a.accumulators().copy())
        sa, pa = a.state(), a.params()
    with g.ResidentTrackingLoop(system, prns, N, M, fs, shifts, re=re, im=im, idle_us=200000, **kw) as n:
        first = n.run(50, keep_all=True)          # every block's accumulators
        assert n.run(nblk - 50, start=50 * N) is None and n.blocks_done == nblk  # only the last block's kept
        sn, pn, last = n.state(), n.params(), n.accumulators()
        assert n.resident.info()["calls"] == nblk
        # argument errors
        lib, h = n._lib, n.resident._h
        buf = np.zeros(2 * prns.size * 3 * M, np.float32)
        args = lambda **o: [o.get("h", h), o.get("nb", 1), o.get("off", 0), N, C.byref(o.get("cfg", n.config)), C.c_void_p(n._state.ctypes.data),
                            C.c_void_p(n._cur.ctypes.data), C.c_void_p(buf.ctypes.data), C.c_void_p(buf.ctypes.data), o.get("stride", 0)]
        assert lib.gat_resident_tracking_run(*args(nb=-1)) == 1  # GAT_ERR_ARG
        assert lib.gat_resident_tracking_run(*args(stride=3)) == 1  # GAT_ERR_ARG
        assert lib.gat_resident_tracking_run(*args(off=1)) == 1  # GAT_ERR_ARG
        assert lib.gat_resident_tracking_run(*args(h=None)) == 1  # GAT_ERR_ARG
        bad = type(n.config).from_buffer_copy(n.config)
        bad.num_taps = 5
        assert lib.gat_resident_tracking_run(*args(cfg=bad)) == 1  # GAT_ERR_ARG
        assert lib.gat_resident_tracking_run(*args(nb=0)) == 0
    assert first.shape == (50, prns.size, 3, M)
    assert np.array_equal(first, np.stack(accs[:50])) and np.array_equal(last, accs[-1])
    assert sa.tobytes() == sn.tobytes() and pa.tobytes() == pn.tobytes()
