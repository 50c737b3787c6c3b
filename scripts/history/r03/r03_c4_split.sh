#!/bin/bash
# Round 3: configs[3] shard (M=16, K=4, B=512) -- workgroups per CU the split planner aims for, with the one-thread-per-
# element second stage; then one bench line per BASELINE shape on the same box.  Output: gpurun_out/r03/c4_split.txt
set -o pipefail
mkdir -p gpurun_out/r03
out=gpurun_out/r03/c4_split.txt
: > $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; l=d['config']['launch']
print('%-14s wgs %-5d kt%d splits %-2d fin %d %.4f ms/launch %.4f ms/step  %s %.3f (hbm %.3f) err %.1e' % ('$1', l['workgroups'], l['channels_per_wg'], l['splits'], l['finalize_launched'], r['kernel_ms_per_launch'], d['ms_per_step'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))"; }
for round in 1 2; do
  for w in 1 2 4 8; do
    GAT_DC_WGS_PER_CU=$w timeout -k 10 200 python bench.py --no-cpu-baseline --baseline-config 3 2>>gpurun_out/r03/c4_split.err | line "c4 wgs/cu=$w" >> $out || exit 1
  done
done
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 20 2>>gpurun_out/r03/c4_split.err | line c2 >> $out
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --num-samples 4000 --num-ants 1 --blocks 16384 2>>gpurun_out/r03/c4_split.err | line c1shape >> $out
timeout -k 10 200 python bench.py --no-cpu-baseline --baseline-config 2 2>>gpurun_out/r03/c4_split.err | line c3 >> $out
timeout -k 10 200 python bench.py --no-cpu-baseline --baseline-config 4 2>>gpurun_out/r03/c4_split.err | line c5 >> $out
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --layout i16 2>>gpurun_out/r03/c4_split.err | line c2_i16 >> $out
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --layout i8 2>>gpurun_out/r03/c4_split.err | line c2_i8 >> $out
cat $out
