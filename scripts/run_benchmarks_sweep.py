#!/usr/bin/env python3
"""Mirror of scripts/run_benchmarks_gpsl1.jl / run_benchmarks_gpsl5.jl of the reference: the same
parameter grid (GPSL1: N = 2^11..2^18, M in {1,4}, L in {3,7}; GPSL5: N = 2^15..2^18, M in {1,4},
L = 3), one satellite channel, 1 ms of signal, sync-inclusive time per call (BenchmarkTools
"Minimum" is what the paper plots, paper/paper.tex:150).  Writes a JSON list (the reference
@tagsave's one JLD2 per point) and prints a table.

usage: python scripts/run_benchmarks_sweep.py [out.json] [seconds-per-point]
"""
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuacceleratedtracking_amd as g  # noqa: E402


def dict_list(params):
    keys = list(params)
    return [dict(zip(keys, vals)) for vals in itertools.product(*[params[k] for k in keys])]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sweep.json"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    grids = [
        {"processor": ["GPU"], "GNSS": ["GPSL1"], "num_samples": [2 ** e for e in range(11, 19)],
         "num_ants": [1, 4], "num_correlators": [3, 7], "algorithm": ["hip_fused", "hip_fused_atomic", "hip_resident"]},
        {"processor": ["GPU"], "GNSS": ["GPSL5"], "num_samples": [2 ** e for e in range(15, 19)],
         "num_ants": [1, 4], "num_correlators": [3], "algorithm": ["hip_fused", "hip_fused_atomic", "hip_resident"]},
    ]
    rows = []
    for grid in grids:
        for d in dict_list(grid):
            try:
                r = g.run_kernel_benchmark(d, seconds=seconds)
            except g.GatError as e:  # a resident correlator serves what is ONE launch otherwise (not: 7 taps at 262 MHz)
                if d["algorithm"] == "hip_resident" and e.status == 4:
                    print(f'{d["GNSS"]} N={d["num_samples"]:7d} M={d["num_ants"]} L={d["num_correlators"]} hip_resident: unsupported ({e})', flush=True)
                    continue
                raise
            fs = d["num_samples"] / 1e-3
            row = {k: r[k] for k in ("GNSS", "num_samples", "num_ants", "num_correlators", "algorithm", "Minimum",
                                     "Median", "Mean", "σ", "Maximum", "os", "CPU_model", "GPU_model", "HIP")}
            row["samples"] = int(len(r["RawTimes"]))
            row["Msamples_per_s_min"] = d["num_samples"] / (r["Minimum"] * 1e-9) / 1e6
            row["real_time_factor"] = 1e-3 / (r["Minimum"] * 1e-9)
            rows.append(row)
            print(f'{d["GNSS"]} N={d["num_samples"]:7d} fs={fs/1e6:8.3f} MHz M={d["num_ants"]} L={d["num_correlators"]} '
                  f'{d["algorithm"]:17s} min {r["Minimum"]/1e3:8.2f} us  median {r["Median"]/1e3:8.2f} us  '
                  f'RTF {row["real_time_factor"]:8.1f}', flush=True)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
