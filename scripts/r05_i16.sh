#!/bin/bash
# Round 5: int16 samples on the split-bf16 kernel with the exact two-term split (5 products per sample) against the float
# path's three terms (8): parity tests, then configs[4] from int16 with both.  Output: gpurun_out/r05/i16_*.txt
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
cd $REPO
PART=${1:-all}
if [ "$PART" = "all" ] || [ "$PART" = "tests" ]; then
  timeout -k 10 600 python -m pytest tests/test_mfma_gpu.py -x -q > $OUT/i16_pytest.log 2>&1; rc=$?; tail -15 $OUT/i16_pytest.log; echo "pytest rc $rc"
  [ $rc -eq 0 ] || exit $rc
fi
if [ "$PART" = "all" ] || [ "$PART" = "time" ]; then
  for terms in 2 3; do
    QARGS="--option mc_i16_terms=$terms" bash scripts/r05_quick.sh i16t$terms c4i16 c4k32i16 m32k32i16 || exit 1
  done
  bash scripts/r05_quick.sh i16f32 c4
  cat $OUT/quick_i16t2.txt $OUT/quick_i16t3.txt $OUT/quick_i16f32.txt > $OUT/i16_two_term_timing.txt
fi
