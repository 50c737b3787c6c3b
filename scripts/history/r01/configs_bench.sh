#!/bin/bash
# Throughput of every BASELINE config that fits one GPU (parity for these shapes is in tests/).
run() { timeout 180 python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print(c['workload']); print('   value %.1f Msamples/s  rtf %.1f  achieved %.1f GB/s frac %.3f  ms/launch %.4f err %.2e launch %s' % (d['value'], d['real_time_factor'], r['achieved'], r['frac'], r['kernel_ms_per_launch'], d['parity_max_rel_err_vs_f64_oracle'], c['launch']))"; }
run --gnss GPSL1 --num-samples 4000  --num-ants 1  --num-taps 3 --channels 1  --blocks 16384      # C1 shape
run --gnss GPSL1 --num-samples 20000 --num-ants 4  --num-taps 3 --channels 1  --blocks 4096       # C2
run --gnss GPSL5 --num-samples 50000 --num-ants 4  --num-taps 5 --channels 12 --blocks 1024       # C3
run --gnss GPSL1 --num-samples 50000 --num-ants 16 --num-taps 3 --channels 4  --blocks 512        # C4 per GPU
run --gnss GPSL1 --num-samples 2000000 --num-ants 64 --num-taps 3 --channels 8 --blocks 2 --block-ms 20   # C5 (8 of 64 channels)
