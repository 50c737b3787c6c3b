#!/bin/bash
# Round 5: BASELINE configs[3] as a whole on ONE GPU (16 antennas x 32 PRNs, 50 MHz, 512 blocks): which tiling
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_constellation_k32.txt; : > $out
for rep in 1 2; do
  bash scripts/r05_quick.sh kt4 c3k32 | tee -a $out
  QARGS="--option dc_kt=2" bash scripts/r05_quick.sh kt2 c3k32 | tee -a $out
  QARGS="--option dc_kt=1" bash scripts/r05_quick.sh kt1 c3k32 | tee -a $out
  QARGS="--matrix-core 3" bash scripts/r05_quick.sh bf16 c3k32 | tee -a $out
  QARGS="--option dc_aw=2 --option dc_aw2=1" bash scripts/r05_quick.sh a2k2 c3k32 | tee -a $out
done
