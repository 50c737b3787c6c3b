#!/usr/bin/env python3
"""Mirror of scripts/benchmark_reduction.jl and scripts/benchmark_textmem.jl of the reference: the reduction grid
(N = 2^11..2^15, M = 4, L = 3, pure / cplx / cplx_multi) and the code-replica grid (N = 2^11..2^18, gmem / textmem),
sync-inclusive time per call.  usage: python scripts/benchmark_reduction.py [out.json] [seconds-per-point]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuacceleratedtracking_amd as g  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/reduction_replica.json"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    res = {"reduction": [], "codereplica": []}
    for n in [2 ** e for e in range(11, 16)]:
        for alg in ("pure", "cplx", "cplx_multi"):
            r = g.run_reduction_benchmark({"num_samples": n, "num_ants": 4, "num_correlators": 3, "algorithm": alg},
                                          seconds=seconds)
            res["reduction"].append(r)
            print(f'reduction N={n:6d} {alg:10s} min {r["Minimum"]/1e3:8.2f} us median {r["Median"]/1e3:8.2f} us', flush=True)
    for n in [2 ** e for e in range(11, 19)]:
        for alg in ("gmem", "textmem"):
            r = g.run_replica_benchmark({"num_samples": n, "algorithm": alg}, seconds=seconds)
            res["codereplica"].append(r)
            print(f'replica   N={n:6d} {alg:10s} min {r["Minimum"]/1e3:8.2f} us median {r["Median"]/1e3:8.2f} us', flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
