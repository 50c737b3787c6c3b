#!/bin/bash
# Round 5: the 2 x 2 antenna tile (two waves of two antennas: option dc_aw2) against the one-wave-of-four tile, same box, two rounds
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_aw2.txt; : > $out
for rep in 1 2; do
  GAT_LIBRARY=$PWD/build/libgat_base.so bash scripts/r05_quick.sh base c2 i8 i16 c1 | tee -a $out
  for v in a4s0 a4s2 a5s0 a5s2 a6s2; do
    QARGS="--option dc_aw2=1" GAT_LIBRARY=$PWD/build/libgat_$v.so bash scripts/r05_quick.sh $v c2 i8 i16 c1 | tee -a $out
  done
done
