#!/bin/bash
for rep in 1 2; do for lay in planar interleaved i16 i8; do
timeout 180 python bench.py --no-cpu-baseline --steps 30 --layout $lay "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lay', 'value %.0f Msamples/s' % d['value'], 'GB/s', r['achieved'], 'frac', r['frac'], 'ms', r['kernel_ms_per_launch'], 'err %.1e' % d['parity_max_rel_err_vs_f64_oracle'])"
done; done
