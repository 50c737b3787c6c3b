#!/usr/bin/env python3
"""The resident correlator while the same device streams: 60 launches of the headline shape (0.4 ms each, every CU busy) are
enqueued on the context's stream, single-block calls are rung into a resident kernel meanwhile.  Its workgroups hold their
slots, so the calls are answered -- how fast, and do they stay right?"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import gpuacceleratedtracking_amd as g  # noqa: E402
from tests.helpers import check_close, make_case, oracle_result  # noqa: E402

print(g.load_library().gat_version().decode())
op, desc, sig, prm_s = g.build_stream("GPSL1", 20000, 4, 3, 1, 4096)
ctx = op.ctx
for N, M in ((2048, 4), (32768, 4)):
    case = make_case(7, N=N, M=M, L=3, K=1, B=1)
    ref = oracle_result(case)
    re = torch.from_numpy(case["re"]).to(ctx.device)
    im = torch.from_numpy(case["im"]).to(ctx.device)
    torch.cuda.current_stream().synchronize()
    d1 = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
    p = case["prm"][0]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    with ctx.open_resident(d1, 1, case["shifts"], case["fs"], idle_us=500000) as res:
        quiet = []
        for _ in range(300):
            t0 = time.perf_counter(); res.correlate(prm); quiet.append(time.perf_counter() - t0)
        ctx.set_codes(op.system.codes)
        for _ in range(60):
            op.launch(desc)          # asynchronous: ~24 ms of streaming work queued
        busy = []
        t_end = time.perf_counter() + 0.020
        while time.perf_counter() < t_end:
            t0 = time.perf_counter(); a, b = res.correlate(prm); busy.append(time.perf_counter() - t0)
            check_close((a + 1j * b)[None], ref)
        torch.cuda.current_stream().synchronize()
        info = res.info()
    q, b_ = np.sort(quiet[50:]) * 1e6, np.sort(busy) * 1e6
    print(f"N={N} M={M} ({info['workgroups']} workgroups, {info['launches']} kernel start(s)): quiet device min {q[0]:.1f} median {q[len(q)//2]:.1f} us | "
          f"while streaming ({len(busy)} calls) min {b_[0]:.1f} median {b_[len(b_)//2]:.1f} p99 {b_[int(len(b_)*0.99)]:.1f} max {b_[-1]:.0f} us", flush=True)
