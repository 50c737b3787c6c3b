#!/usr/bin/env python3
"""Counterpart of scripts/plot_benchmarks.jl (:4-565): processing time vs sampling frequency, log-log, one panel per
(GNSS, antennas, correlators), with the 1 ms real-time line -- from the JSON of scripts/run_benchmarks_sweep.py;
optionally the reduction / code-replica panels (src/plots.jl:1-139) from scripts/benchmark_reduction.py.

The CPU series (the reference's sweep drives "processor" => ["CPU"], scripts/run_benchmarks_gpsl1.jl:6) comes from
`python bench.py --cpu-sweep cpu.json` (the CPU-baseline leg of bench.py: the oracle's 1-thread FP32 4-pass port) and is
drawn when a third JSON is given, so the single-block crossover against the CPU is visible as in
paper/figures/desktopplotscrop.png.

usage: python scripts/plot_benchmarks.py sweep.json out.png [reduction_replica.json out2.png] [--cpu cpu.json]"""
import json
import sys

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402


def sweep_figure(rows, out, cpu_rows=()):
    panels = [("GPSL1", 1, 3), ("GPSL1", 4, 3), ("GPSL1", 4, 7), ("GPSL5", 1, 3), ("GPSL5", 4, 3), ("GPSL1", 1, 7)]
    fig, axes = plt.subplots(2, 3, figsize=(15, 8))
    for ax, (gnss, m, l) in zip(axes.ravel(), panels):
        sel = [r for r in rows if r["GNSS"] == gnss and r["num_ants"] == m and r["num_correlators"] == l]
        for alg, style in (("hip_fused", "o-"), ("hip_fused_atomic", "s--"), ("hip_resident", "d-")):
            pts = sorted((r["num_samples"] / 1e-3, r["Minimum"] * 1e-9) for r in sel if r["algorithm"] == alg)
            if pts:
                ax.plot([p[0] for p in pts], [p[1] for p in pts], style, label=f"MI355X {alg} (minimum)")
        cpu = sorted((r["num_samples"] / 1e-3, r["Minimum"] * 1e-9) for r in cpu_rows
                     if r["GNSS"] == gnss and r["num_ants"] == m and r["num_correlators"] == l)
        if cpu:
            ax.plot([p[0] for p in cpu], [p[1] for p in cpu], "^-", color="tab:red", label="host CPU, 1 thread (FP32 4-pass port, minimum)")
        ax.axhline(1e-3, color="k", lw=0.8)
        ax.text(0.02, 0.93, "1 ms = real time", transform=ax.transAxes, fontsize=8)
        ax.set_xscale("log")
        ax.set_yscale("log")
        ax.set_ylim(1e-6, 1e-2)
        ax.set_xlabel("Sampling Frequency [Hz]")
        ax.set_ylabel("Processing Time [s]")
        ax.set_title(f"{gnss} T=1 ms, M={m}, L={l}")
        ax.grid(True, which="both", lw=0.3)
        ax.legend(fontsize=6, loc="upper left", bbox_to_anchor=(0.0, 0.9))
    fig.suptitle("downconvert + correlate, one 1 ms block per call, sync-inclusive (BenchmarkTools 'Minimum')")
    fig.tight_layout()
    fig.savefig(out, dpi=80)


def aux_figure(res, out):
    fig, (a1, a2) = plt.subplots(1, 2, figsize=(11, 4))
    for alg in ("pure", "cplx", "cplx_multi"):
        pts = sorted((r["num_samples"], r["Minimum"] * 1e-9) for r in res["reduction"] if r["algorithm"] == alg)
        a1.plot([p[0] for p in pts], [p[1] for p in pts], "o-", label=alg)
    a1.set_title("reduction, M=4, L=3 (launch sequences: 24 / 12 / 1)")
    for alg in ("gmem", "textmem"):
        pts = sorted((r["num_samples"], r["Minimum"] * 1e-9) for r in res["codereplica"] if r["algorithm"] == alg)
        a2.plot([p[0] for p in pts], [p[1] for p in pts], "o-", label=alg)
    a2.set_title("code replica (exact vs Float32-coordinate index)")
    for ax in (a1, a2):
        ax.set_xscale("log")
        ax.set_yscale("log")
        ax.set_xlabel("num_samples")
        ax.set_ylabel("Processing Time [s]")
        ax.grid(True, which="both", lw=0.3)
        ax.legend(fontsize=8)
    fig.tight_layout()
    fig.savefig(out, dpi=80)


if __name__ == "__main__":
    argv = list(sys.argv[1:])
    cpu_rows = []
    if "--cpu" in argv:
        i = argv.index("--cpu")
        cpu_rows = json.load(open(argv[i + 1]))
        del argv[i:i + 2]
    sweep_figure(json.load(open(argv[0])), argv[1], cpu_rows)
    if len(argv) > 3:
        aux_figure(json.load(open(argv[2])), argv[3])
