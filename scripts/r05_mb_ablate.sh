#!/bin/bash
# Round 5: what bounds the split-bf16 kernel at configs[4] from int16 (two-term split) -- diagnostic builds that leave a part of
# the work out (-DGAT_ABLATE bits, gat_mfma_bf16.hip: 1 no replica after the first step, 2 no carrier fragments, 4 no sample
# split / store, 8 no MFMA work at all, 16 no sample loads, 32 no step barrier, 256 no fragment fetches, 512 no vector
# preparation of the MFMA operands; results wrong on purpose).  VARIANTS="23 55 ..." SHAPES="c4 ..." select.  Output: gpurun_out/r05/mb_ablate.txt
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO; mkdir -p gpurun_out/r05
: > gpurun_out/r05/mb_ablate.txt
for v in ${VARIANTS:-"" 8 23 7 1 2 4}; do
  lib=$REPO/gpuacceleratedtracking_amd/libgat.so; [ -n "$v" ] && lib=$REPO/build/libgat_mbabl$v.so
  [ -f $lib ] || continue
  for shape in ${SHAPES:-c4i16 c4}; do
    GAT_LIBRARY=$lib bash scripts/r05_quick.sh mbabl$v $shape | sed "s/^/ablate=${v:-none} /" | tee -a gpurun_out/r05/mb_ablate.txt
  done
done
