#!/bin/bash
# careful A/B of two builds on the headline workload (BASELINE configs[1]): alternating runs, kernel ms per launch
for i in 1 2 3 4; do for lib in "" "$@"; do GAT_LIBRARY=$lib timeout 180 python bench.py --no-cpu-baseline --steps 60 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[${lib:-default}] ms %.4f frac %.3f' % (r['kernel_ms_per_launch'], r['frac']))"; done; done
