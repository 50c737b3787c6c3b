#!/bin/bash
# rocprofv3 kernel stats of EXACTLY the default bench command (python3 bench.py), for profiles/
set -u
TAG=${1:-r01d_default}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" > "$OUT/bench_trace.json" 2> "$OUT/trace.log"
python3 "$REPO/scripts/history/r01/summarize_profile.py" "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
tail -1 "$OUT/bench_trace.json" | cut -c1-400
