#!/bin/bash
# SQ counter passes for one bench configuration:  scripts/history/r01/pmc_r.sh <tag> <bench args...>
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/p1" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b1.json" 2> "$OUT/p1.log"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b2.json" 2> "$OUT/p2.log"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dc_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            acc.setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            vg = r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"]
print("kernel dc_kernel: VGPR,SGPR,LDS,grid =", vg)
for k in sorted(acc):
    v = acc[k][-3:] if not k.startswith("_") else acc[k]
    print(f"{k:24s} {sum(v)/len(v):.4g}")
PY
tail -3 "$OUT/p1.log" | head -2
