#!/bin/bash
# Round 5: replica fill by quads on the one-channel int8 tile (option dc_quads) -- same library, option on / off, three rounds
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_i8_quads.txt; : > $out
L=$PWD/build/libgat_q8.so
for rep in 1 2 3; do
  QARGS="--option dc_quads=0" GAT_LIBRARY=$L bash scripts/r05_quick.sh entry i8 | tee -a $out
  QARGS="--option dc_quads=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh quads i8 | tee -a $out
done
