#!/bin/bash
# configs[3] shard: channels per workgroup x prefetch depth
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/c4_ab.txt
: > $out
one() { name=$1; shift
  timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-12s kt%d depth %d splits %-2d lds %-6d %.4f ms  hbm %.3f err %.1e' % ('$name', l['channels_per_wg'], l['prefetch_depth'], l['splits'], l['lds_bytes'], r['kernel_ms_per_launch'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
for round in 1 2; do
  GAT_MC_MODE=0 GAT_DC_DEPTH=1 one c4_kt4_d1 --baseline-config 3
  GAT_MC_MODE=0 one c4_kt4_d2 --baseline-config 3
  GAT_MC_MODE=0 GAT_DC_KT=2 GAT_DC_DEPTH=1 one c4_kt2_d1 --baseline-config 3
  GAT_MC_MODE=0 GAT_DC_KT=2 one c4_kt2_d2 --baseline-config 3
  GAT_MC_MODE=0 GAT_DC_KT=1 one c4_kt1 --baseline-config 3
done
cat $out
