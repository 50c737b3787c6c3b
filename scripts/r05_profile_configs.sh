#!/bin/bash
# Round 5 evidence: rocprofv3 --kernel-trace --stats of `python bench.py [...]` for the default line and every BASELINE shape
# (default protocol: 64 settle + 50 warm-up + 200 timed launches), then the PMC passes of scripts/r05_pmc.sh (FETCH_SIZE /
# WRITE_SIZE in separate runs; SQ sets; clock) per shape.  Output: gpurun_out/r05p/  (copied to profiles/r04/ by hand)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05p; mkdir -p $OUT
export TMPDIR=/tmp
SCR=$(mktemp -d /tmp/r05p.XXXXXX)
cd /tmp
stats() { # tag args...
  tag=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/$tag -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.log
  f=$(find $SCR/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  python3 - $OUT/${tag}_kernel_stats.csv $OUT/$tag.json $tag <<'PY' | tee -a $OUT/summary.txt
import csv, json, sys
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d["roofline"]
cand=[x for x in rows if "dc_kernel" in x["Name"] or "mfma_" in x["Name"]]
top=max(cand or rows,key=lambda x: float(x["TotalDurationNs"]))
fin=[x for x in rows if "finalize" in x["Name"]]
print("%-10s %-62s calls %s avg %.1f us%s | bench: %.4f ms/launch %s frac %.4f (hbm %.4f) value %.1f %s" % (sys.argv[3], top["Name"][:62], top["Calls"], float(top["AverageNs"])/1e3, (" + finalize %.1f us" % (float(fin[0]["AverageNs"])/1e3)) if fin else "", r["kernel_ms_per_launch"], r["bound"], r["frac"], r["hbm_frac"], d["value"], d["unit"]))
PY
}
PART=${1:-all}
if [ "$PART" != "pmc" ]; then
: > $OUT/summary.txt
stats default
stats c1shape --no-single-block  --num-samples 4000 --num-ants 1 --blocks 16384
stats c3 --no-single-block --baseline-config 2
stats c4 --no-single-block --baseline-config 3
stats c5 --no-single-block --baseline-config 4
stats c5_i16 --no-single-block --baseline-config 4 --layout i16
stats c2_i16 --no-single-block  --layout i16
stats c2_i8 --no-single-block  --layout i8
stats c4x32 --num-samples 50000 --num-ants 16 --num-taps 3 --channels 32 --blocks 512 --no-single-block
fi
[ "$PART" = "stats" ] && { cat $OUT/summary.txt; exit 0; }
cd $REPO
for spec in "c2:" "c1shape:--num-samples 4000 --num-ants 1 --blocks 16384" "c3:--baseline-config 2" "c4:--baseline-config 3" "c5:--baseline-config 4" "c5_i16:--baseline-config 4 --layout i16" "c2_i16:--layout i16" "c2_i8:--layout i8" "c4x32:--num-samples 50000 --num-ants 16 --num-taps 3 --channels 32 --blocks 512"; do
  tag=${spec%%:*}; args=${spec#*:}
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $tag "; then continue; fi   # ONLY="c5 c5_i16": a subset (a call is limited to 20 minutes)
  : > gpurun_out/r05/pmc_$tag.txt
  bash scripts/r05_pmc.sh $tag "fetch write sq1 sq2 clk" -- $args > /dev/null 2>&1
  cat gpurun_out/r05/pmc_$tag.txt >> $OUT/summary.txt
done
cat $OUT/summary.txt
