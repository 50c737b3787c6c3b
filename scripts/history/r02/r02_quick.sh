#!/bin/bash
# GPU parity suite + the five bench shapes (one line each) -> gpurun_out/r02h/quick_<tag>.txt
tag=${1:-run}
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/quick_$tag.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $out || { cat $out; exit 1; }
one() { name=$1; shift
  timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-8s mc%d kt%d aw%d splits %-2d bpw %-2d lds %-6d %.4f ms  %s %.3f (hbm %.3f) err %.1e' % ('$name', l['matrix_core'], l['channels_per_wg'], l['ant_tile']//4 if l['ant_tile']>=4 else 1, l['splits'], l['blocks_per_wg'], l['lds_bytes'], r['kernel_ms_per_launch'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
one c2 --steps 100 --warmup 20
one c1shape --num-samples 4000 --num-ants 1 --blocks 16384 --steps 100 --warmup 20
one c3 --baseline-config 2
GAT_MC_MODE=0 one c4vec --baseline-config 3
one c4auto --baseline-config 3
one c2_i16 --layout i16 --steps 100 --warmup 20
one c2_i8 --layout i8 --steps 100 --warmup 20
one c2_il --layout interleaved --steps 100 --warmup 20
GAT_MC_MODE=0 one c5vec --baseline-config 4 --steps 5 --warmup 2 --settle 2
cat $out
