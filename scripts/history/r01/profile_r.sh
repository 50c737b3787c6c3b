#!/bin/bash
# Profile the bench workload with rocprofv3 on the GPU box (run through gpurun).
#   usage: scripts/history/r01/profile_r.sh <round-tag> [bench args...]
# Pass 1: kernel trace + stats.  Pass 2/3: PMC counters (FETCH_SIZE, WRITE_SIZE) in their own
# runs (MI355X_MICROARCH.md, rocprofv3 PMC slots: they do not fit one pass).
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.log"
python3 "$REPO/scripts/history/r01/summarize_profile.py" "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
