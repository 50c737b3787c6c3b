#!/bin/bash
# A/B of library builds in ONE call (same box): scripts/history/r02/r02_ab.sh tag libA libB ...  (lib = "main" or a build/libgat_<name>.so variant)
tag=$1; shift
libs=("$@")
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/ab_$tag.txt
: > $out
one() { lib=$1; name=$2; shift; shift
  L=$PWD/gpuacceleratedtracking_amd/libgat.so; [ $lib != main ] && L=$PWD/build/libgat_$lib.so
  GAT_LIBRARY=$L timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-8s %-8s kt%d splits %-2d bpw %-2d lds %-6d %.4f ms  %s %.3f (hbm %.3f) err %.1e' % ('$lib', '$name', l['channels_per_wg'], l['splits'], l['blocks_per_wg'], l['lds_bytes'], r['kernel_ms_per_launch'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
for round in 1 2; do for lib in "${libs[@]}"; do
  one $lib c2 --steps 100 --warmup 20
  one $lib c1shape --num-samples 4000 --num-ants 1 --blocks 16384 --steps 100 --warmup 20
  one $lib c3 --baseline-config 2
  GAT_MC_MODE=0 one $lib c4 --baseline-config 3
done; done
sort -k2,2 -s $out
