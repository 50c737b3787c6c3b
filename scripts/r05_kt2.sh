#!/bin/bash
# Round 5: two channels on the 2 x 2 antenna tile (AW = 2, MT = 2, KT = 2: half the sample loads per channel) against the product tile
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_kt2.txt; : > $out
for rep in 1 2; do
  GAT_LIBRARY=$PWD/build/libgat_base.so bash scripts/r05_quick.sh base c2 | tee -a $out
  for v in k2w3 k2w4 k2w4s2; do
    QARGS="--option dc_aw2=1" GAT_LIBRARY=$PWD/build/libgat_$v.so bash scripts/r05_quick.sh $v c2 | tee -a $out
    QARGS="--option dc_aw2=1 --option dc_seg=2" GAT_LIBRARY=$PWD/build/libgat_$v.so bash scripts/r05_quick.sh ${v}_seg2 c2 | tee -a $out
  done
done
