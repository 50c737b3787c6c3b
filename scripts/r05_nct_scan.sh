#!/bin/bash
# split-bf16 kernel: column tiles per workgroup (option mc_nct) against the tile count of the shape
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r05
out=gpurun_out/r05/nct_scan_${LAYOUT:-i16}.txt; : > $out
for M in 32 64; do for K in 8 12 16 24 32 48; do
  line="M $M K $K:"
  for nct in 1 2 4; do
    ms=$(python bench.py --no-cpu-baseline --no-single-block --no-read-ceiling --layout ${LAYOUT:-i16} --num-samples 50000 --num-ants $M --channels $K --blocks 64 --matrix-core 3 --option mc_nct=$nct --steps 60 --warmup 20 --settle 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); l=d['config']['launch']; print('%.4f (wg %d thr %d tile %d)' % (d['step_ms']['median'], l['workgroups'], l['threads'], l['ant_tile']))")
    line="$line  nct=$nct $ms"
  done
  echo "$line" | tee -a $out
done; done
