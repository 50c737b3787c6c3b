#!/bin/bash
# Round 5: PMC passes of one bench shape (separate rocprofv3 --pmc runs, --kernel-trace only beside them):
#   scripts/r05_pmc.sh tag [sets] -- bench-args...      sets: any of "fetch write sq1 sq2 clk sq3 sq4 lds lds2 mfma" (default: all but lds)
# GAT_LIBRARY=... selects a variant build.  Output: gpurun_out/r05/pmc_<tag>.txt (one dict per pass, averaged over the
# last launches of the dominant kernel) + the raw csv under gpurun_out/r05/pmc_<tag>_<set>/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
sets="fetch write sq1 sq2 clk"
if [ "$1" != "--" ]; then sets=$1; shift; fi
shift
OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
RAW=$(mktemp -d /tmp/r05pmc.XXXXXX)  # raw counter csv: scratch, only the averaged lines are kept
export TMPDIR=/tmp
cd /tmp
pmc() { # set counters args...
  s=$1; cnt=$2; shift; shift
  timeout -k 10 240 rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $RAW/pmc_${tag}_$s -- python3 $REPO/bench.py --no-cpu-baseline --no-single-block --steps 6 --warmup 40 --settle 64 "$@" > $OUT/pmc_${tag}_$s.json 2> $OUT/pmc_${tag}_$s.log
  python3 - $RAW/pmc_${tag}_$s $tag $s <<'PY' | tee -a $OUT/pmc_$tag.txt
import csv, glob, os, sys
acc={}; name=None
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "dc_kernel" in r["Kernel_Name"] or "mfma_" in r["Kernel_Name"]:
            name=r["Kernel_Name"]
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
            acc.setdefault("_vgpr",[]).append(float(r["VGPR_Count"]))
print(sys.argv[2], sys.argv[3], (name or "?")[:70], {k: round(sum(v[-4:])/len(v[-4:]),1) for k,v in sorted(acc.items())})
PY
}
for s in $sets; do case $s in
  fetch) pmc fetch "FETCH_SIZE" "$@";;
  write) pmc write "WRITE_SIZE" "$@";;
  sq1) pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "$@";;
  sq2) pmc sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" "$@";;
  clk) pmc clk "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_IFETCH" "$@";;
  sq3) pmc sq3 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVE_CYCLES" "$@";;
  sq4) pmc sq4 "SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM" "$@";;
  mfma) pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES" "$@";;
  lds2) pmc lds2 "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "$@";;
  lds) pmc lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "$@";;
esac; done
