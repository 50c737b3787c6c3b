#!/bin/bash
# Round 5: configs[4] from int16 (two-term split) -- rocprofv3 --kernel-trace --stats of the bench command, PMC passes (traffic,
# matrix pipe, LDS), and the vector kernel on the int16 shapes the planner's threshold decides.  Output: gpurun_out/r05/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp
SCR=$(mktemp -d /tmp/r05i.XXXXXX)
cd /tmp
: > $OUT/i16_profile_summary.txt
for spec in "c5:--baseline-config 4" "c5_i16:--baseline-config 4 --layout i16" "c5_i16_three_terms:--baseline-config 4 --layout i16 --option mc_i16_terms=3"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $SCR/$tag -- python3 $REPO/bench.py --no-cpu-baseline --no-single-block $args > $OUT/$tag.json 2> $OUT/$tag.log || exit 1
  f=$(find $SCR/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  python3 - $OUT/${tag}_kernel_stats.csv $OUT/$tag.json $tag <<'PY' | tee -a $OUT/i16_profile_summary.txt
import csv, json, sys
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d["roofline"]
top=max([x for x in rows if "mfma_" in x["Name"] or "dc_kernel" in x["Name"]],key=lambda x: float(x["TotalDurationNs"]))
print("%-20s %-62s calls %s avg %.1f us | bench: %.4f ms/launch %s frac %.4f bf16 issue %s err %.2e" % (sys.argv[3], top["Name"][:62], top["Calls"], float(top["AverageNs"])/1e3, r["kernel_ms_per_launch"], r["bound"], r["frac"], r.get("bf16_issue"), d["parity_max_rel_err_vs_f64_oracle"]))
PY
done
cd $REPO
ONLY="c5 c5_i16" bash scripts/r05_profile_configs.sh pmc > /dev/null 2>&1
cat gpurun_out/r05/pmc_c5.txt gpurun_out/r05/pmc_c5_i16.txt
bash scripts/r05_pmc.sh c5_i16 "mfma lds2" -- --baseline-config 4 --layout i16 || exit 1
