for K in 2 3 4 6; do for mode in 1 0; do GAT_MC_MODE=$mode timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --gnss GPSL1 --num-samples 50000 --num-ants 16 --num-taps 3 --channels $K --blocks 512 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('M=16 K=$K mode=$mode mc=%d ms %.4f' % (c['launch']['matrix_core'], r['kernel_ms_per_launch']))"; done; done
for K in 2 3; do for mode in 1 0; do GAT_MC_MODE=$mode timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --gnss GPSL1 --num-samples 50000 --num-ants 64 --num-taps 3 --channels $K --blocks 128 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('M=64 K=$K mode=$mode mc=%d ms %.4f' % (c['launch']['matrix_core'], r['kernel_ms_per_launch']))"; done; done
