#!/bin/bash
# Round 4: the reference's single-block sweep grids (scripts/run_benchmarks_gpsl1.jl:5-18, run_benchmarks_gpsl5.jl:5-18)
# three ways on one box -- the GPU through the Python host layer (scripts/run_benchmarks_sweep.py), the GPU from native code
# (examples/gat_latency.c, GPS L1 AND GPS L5: what a Julia @benchmark of the shim would see), and the host CPU on the oracle's
# vectorised 4-pass port timed inside C (bench.py --cpu-sweep) -- plotted and tabulated.
set -o pipefail
out=gpurun_out/r04s; mkdir -p $out
timeout -k 10 500 python scripts/run_benchmarks_sweep.py $out/sweep_gpu.json 0.2 > $out/sweep_gpu.txt 2>&1; echo "gpu sweep rc=$?"
timeout -k 10 400 ./build/gat_latency 1500 > $out/sweep_gpu_native.txt 2>&1; echo "native sweep rc=$?"
timeout -k 10 300 python bench.py --cpu-sweep $out/sweep_cpu.json > $out/sweep_cpu.txt 2>&1; echo "cpu sweep rc=$?"
timeout -k 10 120 python scripts/plot_benchmarks.py $out/sweep_gpu.json $out/sweep_single_block.png --cpu $out/sweep_cpu.json; echo "plot rc=$?"
python - <<'PY' | tee $out/r04s_single_block_gpu_vs_cpu.txt
import json
g=json.load(open("gpurun_out/r04s/sweep_gpu.json")); c=json.load(open("gpurun_out/r04s/sweep_cpu.json"))
cpu={(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"]):r["Minimum"] for r in c}
nat={}
lib=""
for line in open("gpurun_out/r04s/sweep_gpu_native.txt"):
    if line.startswith("# one 1 ms"): lib=line.strip().split(";")[-1].strip()
    if line.startswith("#") or "|" not in line: continue
    cols = line.split("|")
    head, host, dev, graph, rest = cols[:5]
    s, n, m, l = head.split()
    resn = float(cols[6].split("/")[0]) if len(cols) > 6 else 0.0  # resident correlator, native (0: not served)
    nat[(s, int(n), int(m), int(l))] = (float(host.split("/")[0]), float(dev.split("/")[0]), float(rest.split()[0]), resn)
print("# one 1 ms block per call, minimum over repeated calls (BenchmarkTools 'Minimum', paper/paper.tex:150); CPU: %s, one thread; %s" % (c[0]["CPU_model"], lib))
resp={(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"]):r["Minimum"] for r in g if r["algorithm"]=="hip_resident"}
print("# launch = gat_downconvert_and_correlate + gat_sync (completion flag); resident = gat_resident_correlate (ring + wait + outputs on the host)")
print("GNSS      N        M L   python: launch / resident us   native: launch host-params / dev-params / resident us   device per call us   CPU us    CPU / launch   CPU / resident")
for r in g:
    if r["algorithm"]!="hip_fused": continue
    k=(r["GNSS"],r["num_samples"],r["num_ants"],r["num_correlators"])
    n=nat.get(k)
    rp=resp.get(k)
    print("%-6s %8d %2d %d   %9.2f / %s       %s   %10.2f   %s" % (k[0],k[1],k[2],k[3],r["Minimum"]/1e3, ("%6.2f" % (rp/1e3)) if rp else "     -",
          ("%8.2f / %8.2f / %s                  %8.2f   " % (n[0], n[1], ("%6.2f" % n[3]) if n[3] else "     -", n[2])) if n else " "*70, cpu[k]/1e3,
          ("%6.2f          %s" % (cpu[k]/1e3/n[1], ("%6.2f" % (cpu[k]/1e3/n[3])) if n[3] else "     -")) if n else ""))
PY
