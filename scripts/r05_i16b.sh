#!/bin/bash
# two-term path after the bank-conflict fixes: parity, timing, LDS counters
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_mfma_gpu.py -x -q > $OUT/i16b_pytest.log 2>&1; rc=$?; tail -5 $OUT/i16b_pytest.log; [ $rc -eq 0 ] || exit $rc
bash scripts/r05_quick.sh i16b c4i16 c4k32i16 m32k32i16 c4 i8k8
QARGS="--option mc_i16_terms=3" bash scripts/r05_quick.sh i16b3 c4i16
bash scripts/r05_pmc.sh c5_i16b "lds2" -- --baseline-config 4 --layout i16
