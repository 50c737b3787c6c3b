#!/bin/bash
# Round 5: where the split-bf16 kernel (int16 samples, two-term split) overtakes the vector kernel: M x K scan at N = 50 000,
# 3 taps, 64 blocks per launch, both kernels forced.  Output: gpurun_out/r05/i16_planner_scan.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r05
out=gpurun_out/r05/i16_planner_scan.txt; : > $out
for M in 16 32 48 64; do for K in 4 8 12 16 24 32; do
  line="M $M K $K:"
  for mc in 0 3; do
    ms=$(python bench.py --no-cpu-baseline --no-single-block --no-read-ceiling --layout ${LAYOUT:-i16} --num-samples 50000 --num-ants $M --channels $K --blocks 64 --matrix-core $mc --steps 60 --warmup 20 --settle 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4f kind %d' % (d['step_ms']['median'], d['config']['launch']['matrix_core']))")
    line="$line  mc=$mc $ms"
  done
  echo "$line" | tee -a $out
done; done
