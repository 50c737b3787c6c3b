#!/usr/bin/env python3
"""One 1 ms block of a whole constellation per call (K channels, what a receiver's loop asks for every millisecond): the
ordinary call + wait against the resident correlator's call, Python host layer, same box.  Prints min / median in us.
Usage: r04_resident_constellation.py [max_workgroups [max_workgroups ...]]  (default: the library's default, 64)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gpuacceleratedtracking_amd as g  # noqa: E402
from tests.helpers import check_close, make_case, oracle_result  # noqa: E402

WGS = [int(a) for a in sys.argv[1:]] or [0]
ctx = g.get_context(own_stream=True)
print(g.load_library().gat_version().decode())
for system, N, M, L, K in (("GPSL1", 20000, 4, 3, 1), ("GPSL1", 20000, 4, 3, 4), ("GPSL1", 20000, 4, 3, 8), ("GPSL1", 20000, 4, 3, 12),
                           ("GPSL1", 4096, 1, 3, 12), ("GPSL5", 50000, 4, 5, 12), ("GPSL1", 50000, 16, 3, 4)):
    case = make_case(K + M, system=system, N=N, M=M, L=L, K=K, B=1)
    ref = oracle_result(case)
    ctx.set_codes(case["codes"])
    re = torch.from_numpy(case["re"]).to(ctx.device)
    im = torch.from_numpy(case["im"]).to(ctx.device)
    o_re = torch.zeros((1, K, L, M), device=ctx.device)
    o_im = torch.zeros_like(o_re)
    p = case["prm"][0]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    pdev = ctx.params_to_device(prm)
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
    call = ctx.prepared_call(desc, pdev, 1, K, case["shifts"], case["fs"], o_re, o_im)
    t_ord = []
    for _ in range(1200):
        t0 = time.perf_counter()
        call()
        ctx.sync()
        t_ord.append(time.perf_counter() - t0)
    info = ctx.last_launch_info()
    o = np.sort(t_ord[200:]) * 1e6
    line = f"{system} N={N} M={M} L={L} K={K}: launch + wait {o[0]:.1f} / {o[len(o) // 2]:.1f} us ({info['workgroups']} workgroups{', second stage' if info['finalize_launched'] else ''}) | resident call"
    for wgs in WGS:
        with ctx.open_resident(desc, K, case["shifts"], case["fs"], idle_us=200000, max_workgroups=wgs) as res:
            t_res = []
            for _ in range(1200):
                t0 = time.perf_counter()
                res.correlate(prm)
                t_res.append(time.perf_counter() - t0)
            a, b = res.correlate(prm)
            check_close((a + 1j * b)[None], ref[0:1])
            rinfo = res.info()
        r = np.sort(t_res[200:]) * 1e6
        line += f" {r[0]:.1f} / {r[len(r) // 2]:.1f} us ({rinfo['workgroups']} workgroups)"
    print(line, flush=True)
