#!/bin/bash
# Round 3: the reference's single-block sweep grid three ways on one box -- the GPU through the Python host layer
# (scripts/run_benchmarks_sweep.py), the GPU from native code (examples/gat_latency.c: what a Julia @benchmark of the shim would
# see), and the host CPU on the oracle's vectorised 4-pass port timed inside C (bench.py --cpu-sweep) -- plotted and tabulated.
set -o pipefail
out=gpurun_out/r03s; mkdir -p $out
timeout -k 10 500 python scripts/run_benchmarks_sweep.py $out/sweep_gpu.json 0.2 > $out/sweep_gpu.txt 2>&1; echo "gpu sweep rc=$?"
timeout -k 10 300 ./build/gat_latency 1500 > $out/sweep_gpu_native.txt 2>&1; echo "native sweep rc=$?"
timeout -k 10 300 python bench.py --cpu-sweep $out/sweep_cpu.json > $out/sweep_cpu.txt 2>&1; echo "cpu sweep rc=$?"
timeout -k 10 120 python scripts/plot_benchmarks.py $out/sweep_gpu.json $out/sweep_single_block.png --cpu $out/sweep_cpu.json; echo "plot rc=$?"
python - <<'PY' | tee gpurun_out/r03s/r03s_single_block_gpu_vs_cpu.txt
import json
g=json.load(open("gpurun_out/r03s/sweep_gpu.json")); c=json.load(open("gpurun_out/r03s/sweep_cpu.json"))
cpu={(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"]):r["Minimum"] for r in c}
nat={}
for line in open("gpurun_out/r03s/sweep_gpu_native.txt"):
    if line.startswith("#") or "|" not in line: continue
    head, host, dev, graph, rest = line.split("|")[:5]
    n, m, l = (int(x) for x in head.split())
    nat[("GPSL1", n, m, l)] = (float(host.split("/")[0]), float(dev.split("/")[0]), float(rest.split()[0]))
print("# one 1 ms block per call, minimum over repeated calls (BenchmarkTools 'Minimum', paper/paper.tex:150); CPU: %s, one thread" % c[0]["CPU_model"])
print("GNSS      N        M L   GPU python us   GPU native host-params / dev-params+flag us   device per call us   CPU us    CPU / GPU native")
for r in g:
    if r["algorithm"]!="hip_fused": continue
    k=(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"])
    n=nat.get(k)
    print("%-6s %8d %2d %d   %9.2f       %s   %10.2f   %s" % (k[0],k[1],k[2],k[3],r["Minimum"]/1e3,
          ("%8.2f / %8.2f                  %8.2f   " % n) if n else " "*58, cpu[k]/1e3, ("%6.2f" % (cpu[k]/1e3/n[1])) if n else ""))
PY
