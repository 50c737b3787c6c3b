#!/bin/bash
# Round 2 evidence: rocprofv3 kernel-trace stats of the default bench and of every BASELINE shape, then PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate runs; SQ counters) -> gpurun_out/r02p/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${GAT_PROFILE_TAG:-r02p}; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
stats() { # tag args...
  tag=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $REPO/bench.py --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.log
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${tag}_kernel_stats.csv
  python3 - $OUT/${tag}_kernel_stats.csv $OUT/$tag.json $tag <<'PY'
import csv, json, sys
rows=list(csv.DictReader(open(sys.argv[1])))
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d["roofline"]
cand=[x for x in rows if "dc_kernel" in x["Name"] or "mfma_" in x["Name"]]
top=max(cand or rows,key=lambda x: float(x["TotalDurationNs"]))
print("%-10s %-60s calls %s avg %.1f us | bench: %.4f ms/launch %s frac %.3f (hbm %.3f)" % (sys.argv[3], top["Name"][:60], top["Calls"], float(top["AverageNs"])/1e3, r["kernel_ms_per_launch"], r["bound"], r["frac"], r["hbm_frac"]))
PY
}
pmc() { # tag counters args...
  tag=$1; cnt=$2; shift; shift
  timeout -k 10 200 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python3 $REPO/bench.py --no-cpu-baseline --steps 4 --warmup 1 --settle 2 "$@" > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.log
  python3 - $OUT/pmc_$tag $tag <<'PY'
import csv, glob, os, sys
acc={}
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "dc_kernel" in r["Kernel_Name"] or "mfma_" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
            acc.setdefault("_vgpr",[]).append(float(r["VGPR_Count"]))
print(sys.argv[2], {k: round(sum(v[-3:])/len(v[-3:]),1) for k,v in sorted(acc.items())})
PY
}
stats default --steps 200 --warmup 50
stats c1shape --steps 50 --warmup 10 --num-samples 4000 --num-ants 1 --blocks 16384
stats c3 --baseline-config 2
stats c4 --baseline-config 3
stats c5 --baseline-config 4
stats c2_i16 --steps 50 --warmup 10 --layout i16
stats c2_i8 --steps 50 --warmup 10 --layout i8
all4() { # tag bench-args...
  t=$1; shift
  pmc ${t}_fetch "FETCH_SIZE" "$@"
  pmc ${t}_write "WRITE_SIZE" "$@"
  pmc ${t}_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "$@"
  pmc ${t}_sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "$@"
  pmc ${t}_clk "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_IFETCH" "$@"  # GRBM_GUI_ACTIVE / 8 XCDs / kernel time = shader clock
}
all4 c2
all4 c3 --baseline-config 2
all4 c4 --baseline-config 3
all4 c5 --baseline-config 4
all4 c1shape --num-samples 4000 --num-ants 1 --blocks 16384
