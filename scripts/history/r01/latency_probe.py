import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpuacceleratedtracking_amd as g
ctx = g.get_context()
for (N, M, L) in ((2048, 4, 3), (16384, 4, 3), (262144, 4, 3)):
    for flags in (0, 1):
        system = g.GPSL1(); fs = N / 1e-3
        sig, _ = g.gen_signal(system, 1, 1500.0, N, num_ants=M)
        sh = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, L), fs, 0.5)
        op = g.StreamCorrelator(system, N, M, 1, 1, sh, fs, flags=flags)
        op.set_params(g.make_params(0, 1.023e6, 1500.0, 0.0, 0.0, shape=(1, 1)))
        d = op.describe(sig.re, sig.im)
        for _ in range(20): op.launch(d)
        ctx.sync()
        # (a) launch + sync per call
        t = []
        for _ in range(300):
            t0 = time.perf_counter_ns(); op.launch(d); ctx.sync(); t.append(time.perf_counter_ns() - t0)
        # (b) host enqueue cost only (no sync), then one sync
        t0 = time.perf_counter_ns()
        for _ in range(300): op.launch(d)
        t_enq = (time.perf_counter_ns() - t0) / 300
        ctx.sync()
        # (c) device time per call via events over 300 back-to-back launches
        ctx.timer_start()
        for _ in range(300): op.launch(d)
        ms = ctx.timer_stop()
        print(f"N={N} M={M} L={L} flags={flags} info={ctx.last_launch_info()['splits']} call+sync min {min(t)/1e3:.1f} us med {np.median(t)/1e3:.1f} us | enqueue {t_enq/1e3:.1f} us | device/call {ms*1e3/300:.1f} us")
