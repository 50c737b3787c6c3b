#!/bin/bash
# Round 5: epilogue invariants kept out of the step loop (registers: c2 128+spills -> 120, one-wave 73 -> 63 ...) against the previous text,
# and the two-channel 2 x 2 tile at four waves per SIMD (option dc_aw2); same box, two rounds
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_ep.txt; : > $out
for rep in 1 2; do
  GAT_LIBRARY=$PWD/build/libgat_old.so bash scripts/r05_quick.sh old c2 c1 c0 c3 i8 i16 | tee -a $out
  GAT_LIBRARY=$PWD/build/libgat_ep.so bash scripts/r05_quick.sh ep c2 c1 c0 c3 i8 i16 | tee -a $out
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$PWD/build/libgat_k2n.so bash scripts/r05_quick.sh k2 c2 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_kt=1" GAT_LIBRARY=$PWD/build/libgat_k2n.so bash scripts/r05_quick.sh a2k1 c2 | tee -a $out
done
