#!/bin/bash
# prefetch depth A/B on one box: GAT_DC_DEPTH=1 (one sample set) vs default (deepest the instance family has)
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/depth_ab.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $out || { cat $out; exit 1; }
one() { name=$1; d=$2; shift; shift
  GAT_DC_DEPTH=$d timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-8s depth<=%d kt%d splits %-2d bpw %-2d lds %-6d %.4f ms  %s %.3f (hbm %.3f) err %.1e' % ('$name', $d, l['channels_per_wg'], l['splits'], l['blocks_per_wg'], l['lds_bytes']*10+l.get('prefetch_depth',0), r['kernel_ms_per_launch'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
for round in 1 2; do for d in 1 4; do
  one c1shape $d --num-samples 4000 --num-ants 1 --blocks 16384 --steps 100 --warmup 20
  GAT_MC_MODE=0 one c4 $d --baseline-config 3
  GAT_MC_MODE=0 GAT_DC_KT=2 one c4kt2 $d --baseline-config 3
  one m1k12 $d --num-samples 20000 --num-ants 1 --channels 12 --blocks 1024 --steps 50 --warmup 10
  one m2 $d --num-samples 20000 --num-ants 2 --blocks 4096 --steps 50 --warmup 10
  one c2 $d --steps 100 --warmup 20
done; done
cat $out
