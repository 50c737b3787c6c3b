#!/usr/bin/env python3
"""Mirror of scripts/code_replica_experiment.jl of the reference: relative code-phase error of the
texture-memory (Float32 normalised-coordinate) code replica against the exact floor/mod replica,
for 1 ms of GPS L1 C/A at N = 2048:32:262144 samples (fs = N / 1 ms).

    err_rel = sum(|rep_exact - rep_f32coord|) / num_samples        (code_replica_experiment.jl:81)

The paper reports min 0 %, mean 0.03 %, median 0.02 %, max 3.17 % (paper/paper.tex:322-331) for
NVIDIA's texture unit; here the Float32-coordinate arithmetic is emulated in a HIP kernel
(gat_gen_code_replica_f32coord).  usage: code_replica_experiment.py [out.json] [step]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gpuacceleratedtracking_amd as g  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/code_replica_experiment.json"
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    system = g.GPSL1(use_gpu=True)
    fc = g.get_code_frequency(system)
    dev = g.get_context().device
    Ns = np.arange(2048, 262144 + 1, step)
    err = np.zeros(Ns.size)
    buf_a = torch.zeros(262144 + 4096, device=dev)
    buf_b = torch.zeros_like(buf_a)
    for i, N in enumerate(Ns):
        N = int(N)
        fs = N / 1e-3
        corr = g.EarlyPromptLateCorrelator(g.NumAnts(1), g.NumAccumulators(3))
        shifts = g.get_correlator_sample_shifts(system, corr, fs, 0.5)
        count = N + int(shifts[-1] - shifts[0])
        g.gen_code_replica(buf_a, system, fc, fs, 0.0, 1, N, shifts, 1)
        g.gen_code_replica(buf_b, system, fc, fs, 0.0, 1, N, shifts, 1, texture_coordinates=True)
        err[i] = float((buf_a[:count] - buf_b[:count]).abs().sum().item()) / N
    pct = 100.0 * err
    stats = {"points": int(Ns.size), "min_pct": float(pct.min()), "mean_pct": float(pct.mean()),
             "median_pct": float(np.median(pct)), "max_pct": float(pct.max()),
             "paper": {"min_pct": 0.0, "mean_pct": 0.03, "median_pct": 0.02, "max_pct": 3.17,
                       "source": "paper/paper.tex:322-331"}}
    print(json.dumps(stats, indent=1))
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as f:
        json.dump({"stats": stats, "N": Ns.tolist(), "err_rel_pct": [round(float(v), 6) for v in pct]}, f)


if __name__ == "__main__":
    main()
