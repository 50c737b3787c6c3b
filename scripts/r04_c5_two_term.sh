#!/bin/bash
# configs[4]: what a 2-term split of the sample operand could buy at most -- diagnostic build (-DGAT_DEV -DGAT_ABLATE=128, wrong
# results on purpose) that runs the arithmetic of two k-slices out of three and splits x into two terms, against the product
# build, alternating on one box.  Headline protocol.
out=gpurun_out/r04f_c5_two_term_time.txt; : > $out
for rep in 1 2 3; do
for lib in gpuacceleratedtracking_amd/libgat.so build/libgat_c5two.so; do
GAT_LIBRARY=$PWD/$lib python bench.py --no-cpu-baseline --baseline-config 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-40s ms %.4f  f32-roof frac %.4f  err %.2e  %s' % (sys.argv[1], r['kernel_ms_per_launch'], r['frac'], d['parity_max_rel_err_vs_f64_oracle'], d.get('libgat','')))" $lib | tee -a $out
done; done
