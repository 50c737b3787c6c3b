#!/bin/bash
# Does the 128-byte alignment of a block's start matter?  configs[3] shard (N = 50 000 floats: odd blocks start 64 bytes
# off a 128-byte line) against N = 49 984 / 50 016 (every block 128-byte aligned); int8 pairs at N = 20 000 (40 000 B:
# odd blocks 64 bytes off) against N = 19 968 / 20 032.  Same protocol as the headline (64 settle + 50 warm-up + 200 timed).
out=gpurun_out/r04a_align_probe.txt; : > $out
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-70s ms/launch %.4f  hbm_frac %.4f' % (' '.join(sys.argv[1:]), r['kernel_ms_per_launch'], r['hbm_frac']))" "$@" | tee -a $out; }
for rep in 1 2; do
for n in 50000 49984 50016; do run --gnss GPSL1 --num-samples $n --num-ants 16 --num-taps 3 --channels 4 --blocks 512; done
for n in 20000 19968 20032; do run --layout i8 --num-samples $n; done
for n in 20000 19968 20032; do run --layout i16 --num-samples $n; done
done
