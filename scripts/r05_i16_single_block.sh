#!/bin/bash
# int16 shapes the round-5 auto rule hands to the split-bf16 kernel, as ONE block per launch (device time, pipelined launches):
# does the rule hold when the launch is short?  Output: gpurun_out/r05/i16_single_block.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r05
out=gpurun_out/r05/i16_single_block.txt; : > $out
for spec in "32 8" "32 16" "64 8" "64 16" "64 32"; do set -- $spec; M=$1; K=$2
 for N in 4000 20000 50000; do
  line="M $M K $K N $N B 1:"
  for mc in 0 3; do
    ms=$(python bench.py --no-cpu-baseline --no-single-block --no-read-ceiling --layout ${LAYOUT:-i16} --num-samples $N --num-ants $M --channels $K --blocks 1 --matrix-core $mc --steps 300 --warmup 50 --settle 50 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); l=d['config']['launch']; print('%.4f ms (kind %d wg %d splits %d)' % (d['step_ms']['median'], l['matrix_core'], l['workgroups'], l['splits']))")
    line="$line  mc=$mc $ms"
  done
  echo "$line" | tee -a $out
 done
done
