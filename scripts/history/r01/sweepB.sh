#!/bin/bash
for rep in 1 2; do for B in 512 768 1024 1280 1536 2048 3072 4096 8192; do
timeout 180 python bench.py --no-cpu-baseline --steps 30 --blocks $B "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('B', $B, 'GB/s', r['achieved'], 'frac', r['frac'], 'ms', r['kernel_ms_per_launch'])"
done; done
