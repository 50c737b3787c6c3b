#!/bin/bash
# Round 5: chip tables of long codes staged as sign bits (1.3 KB instead of 10 KB per GPS L5 PRN), alone and under the two-channel 2 x 2 tile
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_bits.txt; : > $out
L=$PWD/build/libgat_bt.so
for rep in 1 2; do
  QARGS="--option dc_bits=0" GAT_LIBRARY=$L bash scripts/r05_quick.sh int8tab c2 | tee -a $out
  GAT_LIBRARY=$L bash scripts/r05_quick.sh bits c2 | tee -a $out
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh bits_k2 c2 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_seg=6" GAT_LIBRARY=$L bash scripts/r05_quick.sh bits_k2s6 c2 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_seg=4" GAT_LIBRARY=$L bash scripts/r05_quick.sh bits_k2s4 c2 | tee -a $out
  QARGS="--option dc_aw2=1 --option dc_bits=2" GAT_LIBRARY=$L bash scripts/r05_quick.sh bits2_k2 c2l1 c1k8 | tee -a $out
  QARGS="--option dc_aw2=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2 c2l1 c1k8 | tee -a $out
done
