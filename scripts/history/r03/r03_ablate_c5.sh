#!/bin/bash
# Round 3: configs[4] on the split-bf16 kernel with parts switched off (diagnostic builds -DGAT_ABLATE=mask: 1 replica, 2 carrier
# fragments, 4 sample split / store, 8 the MFMA loop, 16 sample loads; results wrong on purpose), at the steady state of the default
# bench protocol -- what would removing the 3x redundant X split buy?  Output: gpurun_out/r03/ablate_c5.txt
mkdir -p gpurun_out/r03
out=gpurun_out/r03/ablate_c5.txt; : > $out
for lib in main abl4 abl7 abl23 abl8; do
  L=$PWD/gpuacceleratedtracking_amd/libgat.so; [ $lib != main ] && L=$PWD/build/libgat_$lib.so
  GAT_LIBRARY=$L timeout -k 10 280 python bench.py --no-cpu-baseline --baseline-config 4 --steps 100 2>>gpurun_out/r03/ablate_c5.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-6s %.4f ms/launch' % ('$lib', r['kernel_ms_per_launch']))" >> $out
done
cat $out
