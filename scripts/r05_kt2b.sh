#!/bin/bash
# Round 5: would two channels per 2 x 2 tile pay if the chip tables were small?  configs[2]'s shape on GPS L1 tables (1 KB instead of 10 KB)
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_kt2_small_tables.txt; : > $out
for rep in 1 2; do
  GAT_LIBRARY=$PWD/build/libgat_ep.so bash scripts/r05_quick.sh ep c2l1 c1k8 i8k8 | tee -a $out
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$PWD/build/libgat_k2n.so bash scripts/r05_quick.sh k2 c2l1 c1k8 i8k8 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_seg=4" GAT_LIBRARY=$PWD/build/libgat_k2n.so bash scripts/r05_quick.sh k2seg4 c2l1 | tee -a $out
done
