#!/bin/bash
# split-bf16 kernel: plain vs non-temporal sample loads -- time (same box, alternating) and HBM traffic
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
ROUNDS=2 bash scripts/r05_ab_libs.sh ntloads "c4 c4i16 m32k64 i8m16k8" plain:gpuacceleratedtracking_amd/libgat.so nt:build/libgat_ntloads.so
bash scripts/r05_pmc.sh c5_plain "fetch" -- --baseline-config 4
GAT_LIBRARY=$REPO/build/libgat_ntloads.so bash scripts/r05_pmc.sh c5_nt "fetch" -- --baseline-config 4
bash scripts/r05_pmc.sh c5_i16_plain "fetch" -- --baseline-config 4 --layout i16
