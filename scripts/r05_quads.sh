#!/bin/bash
# Round 5: the two-channel 2 x 2 tile with the replica fill by quads (int8 and sign-bit tables) against the per-entry walk and the product tile
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_quads.txt; : > $out
L=$PWD/build/libgat_qf.so
for rep in 1 2; do
  GAT_LIBRARY=$L bash scripts/r05_quick.sh base c2 c2l1 c1k8 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_quads=0" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2 c2 c2l1 c1k8 | tee -a $out
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2q c2 c2l1 c1k8 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_bits=0" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2q_i8tab c2 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_bits=2" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2q_bits c2l1 | tee -a $out
done
