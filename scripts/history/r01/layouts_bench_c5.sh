#!/bin/bash
# config-5 shape (64 antennas x 64 channels, 20 ms @ 100 MHz) in every sample format: split-bf16 matrix kernel vs vector kernel
for lay in ${LAYOUTS:-planar interleaved i16 i8}; do for mode in 1 0; do GAT_MC_MODE=$mode timeout 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --gnss GPSL1 --num-samples 2000000 --num-ants 64 --num-taps 3 --channels 64 --blocks 1 --block-ms 20 --layout $lay 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('$lay mode=$mode | rtf %.1f ms %.4f alg GB/s %.0f err %.1e mc=%d' % (d['real_time_factor'], r['kernel_ms_per_launch'], r['achieved'], d['parity_max_rel_err_vs_f64_oracle'], c['launch']['matrix_core']))"; done; done
