#!/bin/bash
# configs[3] shard: cache policy of the sample loads, KT = 2 at three waves per SIMD with splits that fill whole rounds,
# against the default (KT = 4, non-temporal, no split).  Headline protocol.
out=gpurun_out/r04b_c3_probe.txt; : > $out
run() { python bench.py --no-cpu-baseline --baseline-config 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; l=d['config']['launch']
print('%-60s ms %.4f hbm_frac %.4f wgs %d splits %d kt %d fin %d' % (' '.join(sys.argv[1:]) or 'default', r['kernel_ms_per_launch'], r['hbm_frac'], l['workgroups'], l['splits'], l['channels_per_wg'], l['finalize_launched']))" "$@" | tee -a $out; }
for rep in 1 2; do
run
run --option dc_keep_l2=1
run --option dc_kt=2
run --option dc_kt=2 --option dc_wgs_per_cu=12
run --option dc_kt=2 --option dc_wgs_per_cu=6
run --option dc_kt=2 --option dc_keep_l2=0
run --option dc_kt=1
run --option dc_kt=1 --option dc_wgs_per_cu=8
done
