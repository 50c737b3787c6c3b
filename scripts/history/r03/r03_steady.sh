#!/bin/bash
# Round 3: one bench line per BASELINE shape with the default settle (64) + warm-up (50) + 200 timed steps, and the configs[3]
# shard by warm-up length (how long until the device reaches its steady state).  Output: gpurun_out/r03/steady_$1.txt
mkdir -p gpurun_out/r03
out=gpurun_out/r03/steady_${1:-a}.txt
: > $out
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; l=d['config']['launch']
print('%-22s wgs %-5d kt%d splits %-2d fin %d steps %-4d warm %-4d settle %-3d %.4f ms/launch %.4f ms/step  %s %.3f (hbm %.3f) err %.1e' % ('$1', l['workgroups'], l['channels_per_wg'], l['splits'], l['finalize_launched'], d['steps'], d['warmup'], d['settle'], r['kernel_ms_per_launch'], d['ms_per_step'], r['bound'], r['frac'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))"; }
b() { name=$1; shift; timeout -k 10 280 python bench.py --no-cpu-baseline "$@" 2>>gpurun_out/r03/steady.err | line "$name" >> $out; }
for w in 0 10 50 200 1000; do b "c4 settle0 warm=$w" --baseline-config 3 --settle 0 --warmup $w --steps 50; done
b c2 ; b c1shape --num-samples 4000 --num-ants 1 --blocks 16384
b c3 --baseline-config 2; b c4 --baseline-config 3; b c5 --baseline-config 4
b c2_i16 --layout i16; b c2_i8 --layout i8
cat $out
