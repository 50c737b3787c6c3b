#!/bin/bash
# Round 5: where does the two-channel 2 x 2 tile pay?  (planner rule for option dc_aw2 = auto)
mkdir -p gpurun_out/r05; out=gpurun_out/r05/ab_aw2_rule.txt; : > $out
L=$PWD/build/libgat_qf.so
SH="c1k2 c1k3 c1k4 c1k5 c1k7 i16k8 ilk8 m8k4 m12k4 c2i16 lat12 lat4"
GAT_LIBRARY=$L bash scripts/r05_quick.sh base $SH | tee -a $out
QARGS="--option dc_aw2=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2q $SH | tee -a $out
GAT_LIBRARY=$L bash scripts/r05_quick.sh base $SH | tee -a $out
QARGS="--option dc_aw2=1" GAT_LIBRARY=$L bash scripts/r05_quick.sh k2q $SH | tee -a $out
