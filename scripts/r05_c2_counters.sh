#!/bin/bash
# Round 5: counters of the configs[2] kernel (two-channel 2 x 2 tile) and the same shape forced onto the one-wave-of-four tile
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; export TMPDIR=/tmp
mkdir -p gpurun_out/r05
: > gpurun_out/r05/pmc_c2new.txt; : > gpurun_out/r05/pmc_c2old.txt
bash scripts/r05_pmc.sh c2new "fetch write sq1 sq2 clk sq3" -- --baseline-config 2 > /dev/null 2>&1
bash scripts/r05_pmc.sh c2old "fetch sq1 sq2 clk" -- --baseline-config 2 --option dc_aw2=0 > /dev/null 2>&1
cat gpurun_out/r05/pmc_c2new.txt gpurun_out/r05/pmc_c2old.txt
