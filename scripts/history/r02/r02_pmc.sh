#!/bin/bash
# SQ counter passes of a development build on one bench shape: scripts/history/r02/r02_pmc.sh <lib> <tag> [ENV=..]... -- <bench args>
set -u
LIB=$1; TAG=$2; shift; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r02pmc/${LIB}_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp GAT_LIBRARY=$REPO/build/libgat_$LIB.so
for e in "${envs[@]}"; do export "$e"; done
cd /tmp
ARGS="--steps 4 --warmup 1 --settle 2 --no-cpu-baseline $*"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT/p1" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b1.json" 2> "$OUT/p1.log"
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d "$OUT/p2" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b2.json" 2> "$OUT/p2.log"
timeout -k 10 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_IFETCH SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/p3" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b3.json" 2> "$OUT/p3.log"
python3 - "$OUT" "$LIB $TAG" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
acc = {}; vg=None
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dc_kernel" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            acc.setdefault("_dur_ns", []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            vg = r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"]
print("==", sys.argv[2], "VGPR,SGPR,LDS,grid =", vg)
for k in sorted(acc):
    v = acc[k][-3:]
    print(f"  {k:24s} {sum(v)/len(v):.4g}")
PY
