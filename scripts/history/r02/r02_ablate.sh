#!/bin/bash
# Ablation of the fused vector kernel (development builds, -DGAT_DC_ABLATE=bits; results are WRONG by construction):
# 1 = sample loads hit a cache-resident 16 KB, 2 = constant chips (no LDS reads), 4 = no replica fill, 8 = no segment barriers
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/ablate.txt
one() { lib=$1; name=$2; shift; shift
  L=$PWD/gpuacceleratedtracking_amd/libgat.so; [ $lib != base ] && L=$PWD/build/libgat_$lib.so
  GAT_LIBRARY=$L GAT_BENCH_NO_PARITY=1 timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-6s %-6s %.4f ms hbm %.3f err %.1e' % ('$lib', '$name', r['kernel_ms_per_launch'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
: > $out
for lib in base abl15 abl31 abl32 abl63; do
  GAT_MC_MODE=0 one $lib c4 --baseline-config 3
  one $lib c2 --steps 100 --warmup 20
  one $lib c3 --baseline-config 2
  one $lib c1 --num-samples 4000 --num-ants 1 --blocks 16384 --steps 100 --warmup 20
done
cat $out
