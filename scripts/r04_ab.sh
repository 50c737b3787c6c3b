#!/bin/bash
# same-box A/B of two builds over the BASELINE shapes: scripts/r04_ab.sh OUT.txt LIB_A LIB_B [shapes...], alternating, 2 rounds
out=$1; a=$2; b=$3; shift 3
: > gpurun_out/$out
for rep in 1 2; do
  for lib in $a $b; do
    GAT_LIBRARY=$PWD/$lib bash scripts/r04_quick.sh tmp "$@" | sed "s|^tmp |$(basename $lib .so) |" | tee -a gpurun_out/$out
  done
done
