#!/bin/bash
# quick A/B of development builds on the two float 4-antenna shapes (configs[1], configs[2]), two rounds
libs=("$@")
one() { # lib name args...
  lib=$1; name=$2; shift; shift
  GAT_LIBRARY=$PWD/build/libgat_$lib.so GAT_MC_MODE=0 timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-8s %-4s %.4f ms frac %.3f lds %d' % ('$lib', '$name', r['kernel_ms_per_launch'], r['frac'], l['lds_bytes']))"
}
for round in 1 2; do for lib in "${libs[@]}"; do
  one $lib c2 --steps 200 --warmup 50
  one $lib c3 --baseline-config 2
done; done
