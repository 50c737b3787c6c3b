#!/bin/bash
# A/B of two builds of libgat.so on the one-GPU BASELINE shapes: scripts/history/r01/ab_lib.sh <other .so>
run() { for lib in "" "$OTHER"; do GAT_LIBRARY=$lib timeout 180 python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('[${lib:-default}]', c['workload'][:60], '| rtf %.1f  %.1f GB/s frac %.3f ms %.4f err %.1e' % (d['real_time_factor'], r['achieved'], r['frac'], r['kernel_ms_per_launch'], d['parity_max_rel_err_vs_f64_oracle']))"; done; }
OTHER=$1
run --gnss GPSL1 --num-samples 20000 --num-ants 4  --num-taps 3 --channels 1  --blocks 4096       # C2
run --gnss GPSL5 --num-samples 50000 --num-ants 4  --num-taps 5 --channels 12 --blocks 1024       # C3
run --gnss GPSL1 --num-samples 20000 --num-ants 4  --num-taps 3 --channels 8  --blocks 1024       # C2 shape, 8 channels
