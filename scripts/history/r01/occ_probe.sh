#!/bin/bash
# occupancy of bw_probe's pure-read kernels vs dc_kernel (SQ_WAVE_CYCLES is in quad-cycles)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/occ_probe
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -- $REPO/build/bw_probe > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Grid_Size"], r["VGPR_Count"], r["SGPR_Count"])
        acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"]) * 4
    cu = sum(v["SQ_BUSY_CU_CYCLES"]) / len(v["SQ_BUSY_CU_CYCLES"])
    print(k, "waves/CU avg = %.1f" % (wc / cu), "waves", v["SQ_WAVES"][0])
PY
