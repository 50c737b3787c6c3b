#!/bin/bash
# Round 2: how many workgroups per CU should the split planner aim for?  (tail of the last round of resident workgroups
# vs. per-workgroup start-up and the second reduction stage) -- and the box's read ceiling for the same bytes.
mkdir -p gpurun_out/r02h
out=gpurun_out/r02h/split_target.txt
: > $out
{ echo "== read ceiling, configs[1] bytes (8 planes x 4096 blocks x 5000 float4)"; ./build/spp 8 4096 5000
  echo "== read ceiling, configs[3] shard bytes"; ./build/spp; } >> $out 2>&1
one() { name=$1; t=$2; shift; shift
  GAT_DC_WGS_PER_CU=$t timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-6s target %-3d wgs %-6d splits %-3d bpw %d  %.4f ms  %s %.3f (hbm %.3f) err %.1e' % ('$name', $t, l['workgroups'], l['splits'], l['blocks_per_wg'], r['kernel_ms_per_launch'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out
}
for round in 1 2; do
for t in 8 32 64 128; do one c2 $t --steps 100 --warmup 20; done
for t in 8 16 32 64; do GAT_MC_MODE=0 one c4 $t --baseline-config 3; done
for t in 8 32 64; do one c3 $t --baseline-config 2; done
done
for t in 8 32 64; do one c1 $t --num-samples 4000 --num-ants 1 --blocks 16384 --steps 100 --warmup 20; done
cat $out
