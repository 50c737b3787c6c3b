#!/bin/bash
# Round 3: A/B of library builds on ONE box.  scripts/history/r03/r03_ab.sh tag "shapes" libA libB ...
#   lib = "main" (gpuacceleratedtracking_amd/libgat.so) or the NAME of build/libgat_NAME.so
#   shapes = space-separated subset of: c2 c2_i16 c2_i8 c1shape c3 c4 c5
# Every line is the default bench protocol (64 settle + 50 warm-up + 200 timed launches); two rounds, alternating.
tag=$1; shapes=$2; shift; shift
libs=("$@")
mkdir -p gpurun_out/r03
out=gpurun_out/r03/ab_$tag.txt
: > $out
args_of() { case $1 in
  c2) echo "";; c2_i16) echo "--layout i16";; c2_i8) echo "--layout i8";;
  c1shape) echo "--num-samples 4000 --num-ants 1 --blocks 16384";;
  c3) echo "--baseline-config 2";; c4) echo "--baseline-config 3";; c5) echo "--baseline-config 4";; esac; }
one() { lib=$1; name=$2
  L=$PWD/gpuacceleratedtracking_amd/libgat.so; [ $lib != main ] && L=$PWD/build/libgat_$lib.so
  GAT_LIBRARY=$L timeout -k 10 280 python bench.py --no-cpu-baseline $(args_of $name) 2>>gpurun_out/r03/ab_$tag.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; l=d['config']['launch']
print('%-10s %-8s kt%d splits %-2d bpw %-2d depth %d lds %-6d %.4f ms  %s %.3f (hbm %.3f) err %.1e' % ('$lib', '$name', l['channels_per_wg'], l['splits'], l['blocks_per_wg'], l['prefetch_depth'], l['lds_bytes'], r['kernel_ms_per_launch'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
for round in 1 2; do for name in $shapes; do for lib in "${libs[@]}"; do one $lib $name; done; done; done
sort -k2,2 -s $out
