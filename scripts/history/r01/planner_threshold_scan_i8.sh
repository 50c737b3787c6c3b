for M in 16 64; do for K in 2 3 4; do for mode in 3 0; do B=$((8192/M)); GAT_MC_MODE=$mode timeout 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --gnss GPSL1 --num-samples 50000 --num-ants $M --num-taps 3 --channels $K --blocks $B --layout i8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('int8 M=$M K=$K mode=$mode mc=%d ms %.4f' % (c['launch']['matrix_core'], r['kernel_ms_per_launch']))"; done; done; done
