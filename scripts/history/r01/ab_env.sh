#!/bin/bash
# A/B bench with environment variants in one gpurun call: scripts/history/r01/ab_env.sh "<bench args>" "ENV1=.." "ENV2=.." ...
ARGS=$1; shift
for rep in 1 2; do
for ev in "$@"; do
  env $ev timeout 180 python bench.py --no-cpu-baseline --steps 50 $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[$ev]', '$ARGS', 'value', d['value'], 'GB/s', r['achieved'], 'frac', r['frac'], 'ms', r['kernel_ms_per_launch'], 'err', d['parity_max_rel_err_vs_f64_oracle'], d['config']['launch']['lds_bytes'])"
done; done
