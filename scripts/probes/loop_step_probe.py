"""Where a per-block TrackingLoop.step() spends its host time (set_codes / descriptor / correlate call / update call), shape after
shape with native and graph runs in between -- the sequence of scripts/loop_bench.py, in which every other iteration's step()
loop measured 110 us per step instead of 16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
import gpuacceleratedtracking_amd as g
from gpuacceleratedtracking_amd.tracking import _signal_desc
system = g.GPSL1()
side = torch.cuda.Stream(); torch.cuda.set_stream(side)
for (K, M, fs) in [tuple(float(x) if i == 2 else int(x) for i, x in enumerate(s.split(","))) for s in sys.argv[1:]] or ((4, 4, 20e6), (1, 4, 4e6), (4, 4, 20e6), (1, 4, 4e6)):
    N, nblk = int(fs * 1e-3), 400
    prns = np.arange(1, K + 1); dop = np.linspace(-3000, 3000, K)
    prm_sig = g.make_params(prns - 1, 1.023e6, dop, np.linspace(5, 900, K)[None, :], 0.0, shape=(nblk, K))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    mk = lambda: g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop, init_code_phase=np.linspace(5, 900, K))
    a = mk(); ctx = a.ctx
    if os.environ.get('DROP_GRAPHS'): ctx.set_matrix_core(1)  # (kernel-selection calls drop the context's recorded graphs)
    for i in range(20): a.step(re, im, start=i * N)
    ctx.sync()
    t = {"set_codes": 0.0, "desc": 0.0, "correlate": 0.0, "update": 0.0}
    t0a = time.perf_counter()
    for i in range(nblk):
        t0 = time.perf_counter(); ctx.set_codes(system.codes); t1 = time.perf_counter()
        desc = _signal_desc(re, im, N, start=i * N); t2 = time.perf_counter()
        cur, nxt = a._params[a._cur], a._params[1 - a._cur]
        ctx.downconvert_and_correlate(desc, cur, 1, K, a.shifts, a.fs, a.out_re, a.out_im); t3 = time.perf_counter()
        rc = ctx.lib.gat_tracking_update(ctx._h, C.c_void_p(a.out_re.data_ptr()), C.c_void_p(a.out_im.data_ptr()), K, M, C.byref(a.config), C.c_void_p(a._state.data_ptr()), C.c_void_p(cur.data_ptr()), C.c_void_p(nxt.data_ptr())); t4 = time.perf_counter()
        a._cur = 1 - a._cur
        t["set_codes"] += t1 - t0; t["desc"] += t2 - t1; t["correlate"] += t3 - t2; t["update"] += t4 - t3
    t_enq = time.perf_counter() - t0a
    ctx.sync(); tot = time.perf_counter() - t0a
    print(K, M, fs, "per step us:", {k: round(v / nblk * 1e6, 1) for k, v in t.items()}, "enqueue", round(t_enq / nblk * 1e6, 1), "total incl. device", round(tot / nblk * 1e6, 1), {k: v for k, v in ctx.last_launch_info().items() if k in ("workgroups", "splits", "vec", "finalize_launched")},
          "re ptr %x" % re.data_ptr(), flush=True)
    b = mk(); b.run(re, im, 20, keep=False); ctx.sync(); b.run(re, im, nblk, keep=False); ctx.sync()
    c = mk(); c.run(re, im, nblk, keep=False, graph=True); c.run(re, im, nblk, keep=False, graph=True); ctx.sync(); c.run(re, im, nblk, keep=False, graph=True); ctx.sync()
