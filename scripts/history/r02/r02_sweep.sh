#!/bin/bash
# Round 2: the reference's single-block sweep grid on the GPU (scripts/run_benchmarks_sweep.py) and on the GPU box's host
# CPU (bench.py --cpu-sweep: the CPU-baseline leg), plotted together.
set -o pipefail
out=gpurun_out/r02s; mkdir -p $out
timeout -k 10 500 python scripts/run_benchmarks_sweep.py $out/sweep_gpu.json 0.2 > $out/sweep_gpu.txt 2>&1; echo "gpu sweep rc=$?"
timeout -k 10 300 python bench.py --cpu-sweep $out/sweep_cpu.json > $out/sweep_cpu.txt 2>&1; echo "cpu sweep rc=$?"
timeout -k 10 120 python scripts/plot_benchmarks.py $out/sweep_gpu.json $out/sweep_single_block.png --cpu $out/sweep_cpu.json; echo "plot rc=$?"
python - <<'PY'
import json
g=json.load(open("gpurun_out/r02s/sweep_gpu.json")); c=json.load(open("gpurun_out/r02s/sweep_cpu.json"))
cpu={(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"]):r["Minimum"] for r in c}
print("GNSS      N        M L   GPU min us   CPU 1-thread min us   GPU/CPU")
for r in g:
    if r["algorithm"]!="hip_fused": continue
    k=(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"])
    print("%-6s %8d %2d %d   %9.2f   %12.2f        %6.2f" % (k[0],k[1],k[2],k[3],r["Minimum"]/1e3,cpu[k]/1e3,cpu[k]/r["Minimum"]))
PY
