#!/bin/bash
# one PMC pass with a caller-chosen counter list: scripts/history/r01/pmc_generic.sh <tag> "<counters>" <bench args...>
set -u
TAG=$1; CNT=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcg_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/p" -- python3 "$REPO/bench.py" --steps 4 --warmup 1 --no-cpu-baseline "$@" > "$OUT/b.json" 2> "$OUT/p.log"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dc_kernel" in r["Kernel_Name"] or "mfma_" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            acc.setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc):
    v = acc[k][-3:] if not k.startswith("_") else acc[k]
    print(f"{k:24s} {sum(v)/len(v):.5g}")
PY
grep -i "error\|invalid\|not" "$OUT/p.log" | head -3
