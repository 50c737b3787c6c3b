#!/bin/bash
# crossover of the split-bf16 kernel against the vector kernel in the LENGTH of the launch (blocks per launch at N = 50 000)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r05
out=gpurun_out/r05/blocks_scan_${LAYOUT:-i16}.txt; : > $out
for spec in ${SPECS:-"32:8" "64:16" "64:32"}; do M=${spec%%:*}; K=${spec#*:}
 for B in 2 4 8 16 32; do
  line="M $M K $K N 50000 B $B:"
  for mc in 0 3; do
    ms=$(python bench.py --no-cpu-baseline --no-single-block --no-read-ceiling --layout ${LAYOUT:-i16} --num-samples 50000 --num-ants $M --channels $K --blocks $B --matrix-core $mc --steps 200 --warmup 50 --settle 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); l=d['config']['launch']; print('%.4f ms (kind %d wg %d splits %d)' % (d['step_ms']['median'], l['matrix_core'], l['workgroups'], l['splits']))")
    line="$line  mc=$mc $ms"
  done
  echo "$line" | tee -a $out
 done
done
