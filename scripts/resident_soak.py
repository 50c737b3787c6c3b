#!/usr/bin/env python3
"""Soak of the resident correlator (gat_resident_*): tens of thousands of calls at random distances around the kernel's idle
limit, lifetime and call budget, several geometries -- every call must return the bits of the first one, none may hang, and
the kernel must have left and come back many times; both places of the doorbell (device memory behind the BAR, pinned host
memory).  usage: python scripts/resident_soak.py [calls-per-geometry-and-doorbell]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gpuacceleratedtracking_amd as g  # noqa: E402
from tests.helpers import check_close, make_case, oracle_result  # noqa: E402


def main():
    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    ctx = g.get_context()
    print(g.load_library().gat_version().decode())
    for N, M, L, K, pollers, bell in [(n, m, l, k, p, b) for b in (2, 1) for n, m, l, k, p in
                                      ((2048, 4, 3, 1, 0), (16384, 4, 3, 1, 0), (16384, 1, 7, 2, 1), (65536, 4, 3, 1, 0), (262144, 1, 3, 1, 0), (20000, 4, 3, 12, 0))]:
        case = make_case(N % 1000 + M, N=N, M=M, L=L, K=K, B=1)
        ref = oracle_result(case)
        ctx.set_codes(case["codes"])
        re = torch.from_numpy(case["re"]).to(ctx.device)
        im = torch.from_numpy(case["im"]).to(ctx.device)
        torch.cuda.synchronize()
        desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
        p = case["prm"][0]
        prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
        rng = np.random.default_rng(N + M)
        t_all = time.perf_counter()
        with ctx.open_resident(desc, K, case["shifts"], case["fs"], idle_us=250, life_ms=20, max_calls=501, host_pollers=pollers, doorbell=bell) as res:
            first = tuple(a.copy() for a in res.correlate(prm))
            check_close((first[0] + 1j * first[1])[None], ref)
            lat = []
            for i in range(calls):
                gap = rng.choice([0.0, rng.uniform(0, 500e-6), rng.uniform(0, 30e-3) if i % 997 == 0 else 0.0])
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < gap:
                    pass
                t0 = time.perf_counter()
                r, i_ = res.correlate(prm)
                lat.append(time.perf_counter() - t0)
                assert np.array_equal(r, first[0]) and np.array_equal(i_, first[1]), f"call {i} differs"
            info = res.info()
        lat = np.sort(np.asarray(lat)) * 1e6
        print(f"N={N} M={M} L={L} K={K} workgroups {info['workgroups']} ({'doorbell in device memory' if bell == 2 else 'host doorbell, forwarded' if info['workgroups'] > (pollers or 20) else 'host doorbell, polled directly'}): "
              f"{calls} calls bit-identical, kernel started {info['launches']} times, call min {lat[0]:.1f} median {lat[len(lat) // 2]:.1f} "
              f"p99 {lat[int(len(lat) * 0.99)]:.1f} max {lat[-1]:.0f} us, {time.perf_counter() - t_all:.1f} s", flush=True)


if __name__ == "__main__":
    main()
