#!/bin/bash
# Round 2: parity of the re-tiled vector kernel, then A/B of its tilings on the BASELINE shapes (one gpurun call).
set -o pipefail
out=gpurun_out/r02b; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
rc=$?; tail -5 $out/pytest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
run() { # name, env..., -- args
  name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 240 python bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1]); r=d["roofline"]; l=d["config"]["launch"]
    print("%-28s %9.4f ms  %-10s frac %.3f hbm %.3f  err %.1e  wg %d kt %d bpw %d at %d lds %d mc %d" % ("$name", r["kernel_ms_per_launch"], r["bound"], r["frac"], r["hbm_frac"], d["parity_max_rel_err_vs_f64_oracle"], l["workgroups"], l["channels_per_wg"], l["blocks_per_wg"], l["ant_tile"], l["lds_bytes"], l["matrix_core"]))
except Exception as e: print("$name FAILED", e)
PY
}
run c2_default -- --steps 100 --warmup 20
run c1_bpw16 -- --steps 50 --warmup 10 --num-samples 4000 --num-ants 1 --blocks 16384
run c1_bpw1 GAT_DC_BPW=1 -- --steps 50 --warmup 10 --num-samples 4000 --num-ants 1 --blocks 16384
run c3_kt1 GAT_MC_MODE=0 GAT_DC_KT=1 -- --baseline-config 2
run c3_kt2 GAT_MC_MODE=0 GAT_DC_KT=2 -- --baseline-config 2
run c4_vec_aw4_kt4 GAT_MC_MODE=0 -- --baseline-config 3
run c4_vec_aw4_kt2 GAT_MC_MODE=0 GAT_DC_KT=2 -- --baseline-config 3
run c4_vec_aw4_kt1 GAT_MC_MODE=0 GAT_DC_KT=1 -- --baseline-config 3
run c4_vec_aw1_kt1 GAT_MC_MODE=0 GAT_DC_KT=1 GAT_DC_AW=1 -- --baseline-config 3
run c4_auto -- --baseline-config 3
run c5_auto -- --baseline-config 4
run c2_i16 -- --steps 50 --warmup 10 --layout i16
run c2_i8 -- --steps 50 --warmup 10 --layout i8
run c2_il -- --steps 50 --warmup 10 --layout interleaved
run c2_k8_kt4 -- --steps 30 --warmup 5 --channels 8 --blocks 1024
run c2_k8_kt2 GAT_DC_KT=2 -- --steps 30 --warmup 5 --channels 8 --blocks 1024
run c2_k8_kt1 GAT_DC_KT=1 -- --steps 30 --warmup 5 --channels 8 --blocks 1024
run m1_k12_kt4 -- --steps 30 --warmup 5 --num-ants 1 --channels 12 --blocks 1024
run m1_k12_kt1 GAT_DC_KT=1 -- --steps 30 --warmup 5 --num-ants 1 --channels 12 --blocks 1024
