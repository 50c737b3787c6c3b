#!/bin/bash
# A/B bench of several libgat builds in one gpurun call: scripts/history/r01/ab.sh "<bench args>" lib1 lib2 ...
ARGS=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  GAT_LIBRARY=$lib timeout 180 python bench.py --no-cpu-baseline --steps 50 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', '$ARGS', 'value', d['value'], 'GB/s', r['achieved'], 'frac', r['frac'], 'ms', r['kernel_ms_per_launch'], 'err', d['parity_max_rel_err_vs_f64_oracle'])"
done; done
