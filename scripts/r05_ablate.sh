#!/bin/bash
# Round 5: what bounds a shape's vector kernel -- the diagnostic builds (-DGAT_DC_ABLATE=bits: wrong results on purpose) of
# gat_dc_body.inc timed against the product text on one box, two alternating rounds:
#   scripts/r05_ablate.sh OUT.txt SHAPE name:lib [name:lib ...]       SHAPE: c1 c0 c2 c3 i8 i16 (scripts/r04_quick.sh)
# bits: 1 sample loads hit 16 KB (no HBM stream), 2 no chip reads from LDS, 4 no replica fill, 8 no segment barriers,
#       16 no sample refill loads, 32 no per-antenna scheduling fence
out=gpurun_out/$1; shape=$2; shift 2
mkdir -p "$(dirname $out)"; : > $out
for rep in 1 2; do
  for spec in "$@"; do
    name=${spec%%:*}; lib=${spec#*:}
    GAT_LIBRARY=$PWD/$lib timeout -k 10 240 bash scripts/r04_quick.sh tmp $shape | sed "s|^tmp |$name |" | tee -a $out
  done
done
