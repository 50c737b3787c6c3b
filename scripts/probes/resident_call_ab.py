#!/usr/bin/env python3
"""Resident call latency over a few shapes, Python host layer, min / median in us -- for same-box comparisons of library
variants: GAT_LIBRARY=build/libgat_<variant>.so python scripts/probes/resident_call_ab.py [doorbell]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import gpuacceleratedtracking_amd as g  # noqa: E402
from tests.helpers import check_close, make_case, oracle_result  # noqa: E402

bell = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = g.get_context(own_stream=True)
out = [g.load_library().gat_version().decode()]
for N, M, L, K in ((2048, 1, 3, 1), (2048, 4, 3, 1), (16384, 4, 3, 1), (262144, 4, 3, 1), (20000, 4, 3, 12)):
    case = make_case(N % 997 + K, N=N, M=M, L=L, K=K, B=1)
    ref = oracle_result(case)
    ctx.set_codes(case["codes"])
    re = torch.from_numpy(case["re"]).to(ctx.device)
    im = torch.from_numpy(case["im"]).to(ctx.device)
    p = case["prm"][0]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    torch.cuda.synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
    with ctx.open_resident(desc, K, case["shifts"], case["fs"], idle_us=200000, doorbell=bell) as res:
        t = []
        for _ in range(3000):
            t0 = time.perf_counter()
            res.correlate(prm)
            t.append(time.perf_counter() - t0)
        a, b = res.correlate(prm)
        check_close((a + 1j * b)[None], ref[0:1])
        w = res.info()["workgroups"]
    t = np.sort(t[500:]) * 1e6
    out.append(f"N={N} M={M} K={K} ({w} wgs): {t[0]:.2f} / {t[len(t) // 2]:.2f}")
print(" | ".join(out), flush=True)
