#!/bin/bash
# A/B of development builds on the configs[3] shard shape (16 antennas, 4 PRNs) with 2 and 4 channels per workgroup
libs=("$@")
one() { lib=$1; name=$2; kt=$3; shift; shift; shift
  GAT_LIBRARY=$PWD/build/libgat_$lib.so GAT_MC_MODE=0 GAT_DC_KT=$kt timeout -k 10 240 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['config']['launch']
print('%-8s %-6s kt%d %.4f ms hbm %.3f err %.1e lds %d splits %d' % ('$lib', '$name', l['channels_per_wg'], r['kernel_ms_per_launch'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle'], l['lds_bytes'], l['splits']))"
}
for round in 1 2; do for lib in "${libs[@]}"; do
  one $lib c4 4 --baseline-config 3
  one $lib c4 2 --baseline-config 3
done; done
