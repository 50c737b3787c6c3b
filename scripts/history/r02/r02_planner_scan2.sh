one() { # M K B layout mode
  GAT_MC_MODE=$5 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --settle 4 --gnss GPSL1 --num-samples 50000 --num-ants $1 --num-taps 3 --channels $2 --blocks $3 --layout $4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('%-7s M=%-3d K=%-3d mode=%d mc=%d kt=%d  %.4f ms' % ('$4', $1, $2, $5, c['launch']['matrix_core'], c['launch']['channels_per_wg'], r['kernel_ms_per_launch']))"
}
for mode in 3 0; do one 64 32 16 planar $mode; done
for mode in 3 0; do one 32 32 64 planar $mode; done
for mode in 3 0; do one 32 64 32 planar $mode; done
for mode in 3 0; do one 64 24 16 planar $mode; done
for mode in 3 0; do one 128 16 16 planar $mode; done
for mode in 3 0; do one 64 32 16 i16 $mode; done
