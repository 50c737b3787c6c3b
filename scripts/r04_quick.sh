#!/bin/bash
# quick A/B line per shape: scripts/r04_quick.sh TAG [shapes...]  (GAT_LIBRARY selects the build)
tag=$1; shift
out=gpurun_out/r04_quick_$tag.txt; : > $out
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-single-block "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; l=d['config']['launch']
print('%-8s %-10s ms %.4f frac %.4f (%s) hbm_frac %.4f err %.2e  %s' % (sys.argv[1], sys.argv[2], r['kernel_ms_per_launch'], r['frac'], r['bound'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle'], d.get('libgat','')))" $tag $name | tee -a $out; }
for s in "$@"; do
case $s in
c1) run c1 ;;
c0) run c0 --num-samples 4000 --num-ants 1 --blocks 16384 ;;
c2) run c2 --baseline-config 2 ;;
c3) run c3 --baseline-config 3 ;;
i8) run i8 --layout i8 ;;
i16) run i16 --layout i16 ;;
il) run il --layout interleaved ;;
esac
done
