#!/bin/bash
# Round 2: where does the split-bf16 matrix kernel (GAT_MC_MODE=3) beat the re-tiled vector kernel (GAT_MC_MODE=0)?
# N = 50 000, 3 taps; f32 planar and int8 pairs.
one() { # M K B layout mode
  GAT_MC_MODE=$5 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --gnss GPSL1 --num-samples 50000 --num-ants $1 --num-taps 3 --channels $2 --blocks $3 --layout $4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('%-7s M=%-3d K=%-3d mode=%d mc=%d kt=%d  %.4f ms' % ('$4', $1, $2, $5, c['launch']['matrix_core'], c['launch']['channels_per_wg'], r['kernel_ms_per_launch']))"
}
for lay in planar i8; do
for K in 4 6 8 12 16 32; do B=512; [ $K -ge 12 ] && B=128; for mode in 3 0; do one 16 $K $B $lay $mode; done; done
for K in 4 8 16; do for mode in 3 0; do one 32 $K 128 $lay $mode; done; done
for K in 4 8 16 64; do B=64; [ $K -ge 16 ] && B=16; for mode in 3 0; do one 64 $K $B $lay $mode; done; done
done
