#!/bin/bash
# A/B of development builds of libgat (build/libgat_<name>.so) on the BASELINE shapes: scripts/history/r02/r02_variants.sh name...
set -o pipefail
out=gpurun_out/r02v; mkdir -p $out
run() { # lib, name, env..., -- args
  lib=$1; name=$2; shift; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env GAT_LIBRARY=$PWD/build/libgat_$lib.so "${envs[@]}" timeout -k 10 240 python bench.py --no-cpu-baseline "$@" > $out/${lib}_$name.json 2> $out/${lib}_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/${lib}_$name.json").read().strip().splitlines()[-1]); r=d["roofline"]; l=d["config"]["launch"]
    print("%-6s %-16s %9.4f ms  %-10s frac %.3f hbm %.3f  err %.1e  wg %d kt %d bpw %d at %d lds %d mc %d" % ("$lib", "$name", r["kernel_ms_per_launch"], r["bound"], r["frac"], r["hbm_frac"], d["parity_max_rel_err_vs_f64_oracle"], l["workgroups"], l["channels_per_wg"], l["blocks_per_wg"], l["ant_tile"], l["lds_bytes"], l["matrix_core"]))
except Exception as e: print("$lib $name FAILED", e, open("$out/${lib}_$name.err").read()[-300:])
PY
}
for lib in "$@"; do
  run $lib c2 -- --steps 100 --warmup 20
  run $lib c1 -- --steps 50 --warmup 10 --num-samples 4000 --num-ants 1 --blocks 16384
  run $lib c3_kt1 GAT_MC_MODE=0 GAT_DC_KT=1 -- --baseline-config 2
  run $lib c4_aw4_kt1 GAT_MC_MODE=0 GAT_DC_KT=1 -- --baseline-config 3
  run $lib c4_aw4_kt2 GAT_MC_MODE=0 GAT_DC_KT=2 -- --baseline-config 3
  run $lib c3_kt2 GAT_MC_MODE=0 GAT_DC_KT=2 -- --baseline-config 2
  run $lib c4_aw4_kt4 GAT_MC_MODE=0 GAT_DC_KT=4 -- --baseline-config 3
  run $lib c2k8_kt1 GAT_DC_KT=1 -- --steps 30 --warmup 5 --channels 8 --blocks 1024
  run $lib c2k8_kt2 GAT_DC_KT=2 -- --steps 30 --warmup 5 --channels 8 --blocks 1024
done
