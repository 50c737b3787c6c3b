"""Closed tracking loop per 1 ms block: per-block step() calls from Python vs one native gat_tracking_run call (both never
leave the device), and the loop with the HOST in it -- a resident correlator's call + gat_tracking_update_host per block, as
the reference's receiver closes its loops on the CPU (ResidentTrackingLoop; K <= 16).
usage: python scripts/loop_bench.py [K,M,fs ...]
(The per-block step() column of every third shape in one process reads ~110 us instead of ~16: the time sits inside
hipLaunchKernel on the host -- 45 us per launch for that whole 400-step loop, device kernels unchanged at 7 us, whatever the shape
and whether the recorded graphs are dropped or not: scripts/probes/loop_step_probe.py, rocprofv3 --hip-trace.  The native run,
the graph replay and the resident correlator do not go through that path.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpuacceleratedtracking_amd as g

system = g.GPSL1()
side = torch.cuda.Stream()
torch.cuda.set_stream(side)  # a non-default stream: hipGraph capture is not allowed on the legacy default stream
for (K, M, fs) in [tuple(float(x) if i == 2 else int(x) for i, x in enumerate(s.split(","))) for s in sys.argv[1:]] or ((1, 4, 4e6), (4, 4, 20e6), (12, 4, 20e6), (4, 16, 50e6)):
    N, nblk = int(fs * 1e-3), 400
    prns = np.arange(1, K + 1)
    dop = np.linspace(-3000, 3000, K)
    prm_sig = g.make_params(prns - 1, 1.023e6, dop, np.linspace(5, 900, K)[None, :], 0.0, shape=(nblk, K))
    re, im = g.gen_signal_stream(system, prm_sig, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, 3), fs, 0.5)
    mk = lambda: g.TrackingLoop(system, prns, N, M, fs, shifts, init_carrier_doppler=dop, init_code_phase=np.linspace(5, 900, K))
    a = mk(); ctx = a.ctx
    for i in range(20): a.step(re, im, start=i * N)
    ctx.sync(); t0 = time.perf_counter()
    for i in range(nblk): a.step(re, im, start=i * N)
    ctx.sync(); t_step = (time.perf_counter() - t0) / nblk
    b = mk(); b.run(re, im, 20, keep=False); ctx.sync(); t0 = time.perf_counter()
    b.run(re, im, nblk, keep=False)
    ctx.sync(); t_run = (time.perf_counter() - t0) / nblk
    c = mk(); c.run(re, im, nblk, keep=False, graph=True); c.run(re, im, nblk, keep=False, graph=True); ctx.sync(); t0 = time.perf_counter()
    c.run(re, im, nblk, keep=False, graph=True)
    ctx.sync(); t_graph = (time.perf_counter() - t0) / nblk
    print(f"   hipGraph replay {t_graph*1e6:.1f} us (RTF {1e-3/t_graph:.0f})")
    print(f"K={K} M={M} fs={fs/1e6:g} MHz: per-block step() {t_step*1e6:.1f} us (RTF {1e-3/t_step:.0f}) | native run {t_run*1e6:.1f} us (RTF {1e-3/t_run:.0f})")
    if K <= 16 and not os.environ.get("GAT_LOOP_BENCH_NO_RESIDENT"):
        torch.cuda.synchronize()
        with g.ResidentTrackingLoop(system, prns, N, M, fs, shifts, re=re, im=im, init_carrier_doppler=dop, init_code_phase=np.linspace(5, 900, K),
                                    idle_us=200000) as h:
            for b in range(20):
                h.step(b * N)
            t0 = time.perf_counter()
            for b in range(20, nblk):
                h.step(b * N)
            t_host = (time.perf_counter() - t0) / (nblk - 20)
            h.run(20); t0 = time.perf_counter(); h.run(nblk - 20, start=20 * N); t_nat = (time.perf_counter() - t0) / (nblk - 20)
            print(f"   host-closed loop through the resident correlator ({h.resident.info()['workgroups']} workgroups; every block's accumulators and parameters are "
                  f"on the host): stepped from Python {t_host*1e6:.1f} us per block (RTF {1e-3/t_host:.0f}) | native run (gat_resident_tracking_run) "
                  f"{t_nat*1e6:.1f} us (RTF {1e-3/t_nat:.0f})")
