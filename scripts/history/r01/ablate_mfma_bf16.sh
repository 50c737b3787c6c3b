#!/bin/bash
# kernel time of the config-5 shape with parts of the split-bf16 kernel switched off (diagnostic builds
# -DGAT_ABLATE=mask: 1 replica, 2 carrier fragments, 4 sample split/store, 8 MFMA loop; results are wrong on purpose)
for lay in planar i8; do for lib in ${LIBS:-"" build/libgat_abl1.so build/libgat_abl2.so build/libgat_abl4.so build/libgat_abl7.so build/libgat_abl8.so}; do
GAT_LIBRARY=$lib timeout 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --gnss GPSL1 --num-samples 2000000 --num-ants 64 --num-taps 3 --channels 64 --blocks 1 --block-ms 20 --layout $lay 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lay ${lib:-full} | ms %.4f' % r['kernel_ms_per_launch'])"; done; done
