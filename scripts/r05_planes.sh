#!/bin/bash
# Round 5: two-plane replica layout (conflict-free 8-byte chip reads of the two-sample passes) against the linear layout (libgat_qf.so)
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_planes.txt; : > $out
for rep in 1 2; do
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$PWD/build/libgat_qf.so bash scripts/r05_quick.sh linear c2 c2i16 | tee -a $out
  GAT_LIBRARY=$PWD/build/libgat_pl.so bash scripts/r05_quick.sh planes c2 c2i16 c1k8 | tee -a $out
  QARGS="--option dc_aw2=0" GAT_LIBRARY=$PWD/build/libgat_qf.so bash scripts/r05_quick.sh linear_k1 c2 | tee -a $out
  QARGS="--option dc_aw2=0" GAT_LIBRARY=$PWD/build/libgat_pl.so bash scripts/r05_quick.sh planes_k1 c2 | tee -a $out
done
