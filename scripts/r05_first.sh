#!/bin/bash
# Round 5, first GPU call: the whole -m gpu suite, the default bench line + the constellation leg, the ablation matrix of
# configs[2] and of the int8 stream (what bounds them), extended SQ counters of configs[2].  Output: gpurun_out/r05/
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r05; mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
PART=${1:-all}
if [ "$PART" = "all" ] || [ "$PART" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
if [ "$PART" = "all" ] || [ "$PART" = "bench" ]; then
  timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.log; echo "bench rc $?"
  timeout -k 10 400 python bench.py --constellation --no-cpu-baseline --no-single-block > $OUT/bench_constellation.json 2> $OUT/bench_constellation.log; echo "constellation rc $?"
  python - <<'PY'
import json
for f in ("bench_default", "bench_constellation"):
    try:
        d = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "no line:", e); continue
    r = d["roofline"]
    print(f, "value", d["value"], "ms/step", d["ms_per_step"], "step_ms", d.get("step_ms"), "frac", r["frac"], "frac_mean", r.get("frac_mean"),
          "ceiling", r.get("read_ceiling_GBps"), "frac_of_ceiling", r.get("frac_of_ceiling"), "terms", r["terms_ms"])
    c = d.get("constellation_config3")
    if c:
        print("  constellation:", c["workload"], "| ms/step", c["ms_per_step"], "RTF", c["real_time_factor"], "step_ms", c["step_ms"],
              "launch", c["launch"], "frac", c["roofline"]["frac"], c["roofline"]["bound"], "hbm_frac", c["roofline"]["hbm_frac"], "err", c["parity_max_rel_err_vs_f64_oracle"])
PY
fi
if [ "$PART" = "all" ] || [ "$PART" = "ablate" ]; then
  LIBS="base:build/libgat_base.so abl1:build/libgat_abl1.so abl2:build/libgat_abl2.so abl4:build/libgat_abl4.so abl8:build/libgat_abl8.so abl12:build/libgat_abl12.so abl16:build/libgat_abl16.so abl13:build/libgat_abl13.so abl29:build/libgat_abl29.so"
  bash scripts/r05_ablate.sh r05/ablate_c2.txt c2 $LIBS
  bash scripts/r05_ablate.sh r05/ablate_i8.txt i8 $LIBS
fi
if [ "$PART" = "all" ] || [ "$PART" = "pmc" ]; then
  rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
  export GAT_LIBRARY=$REPO/build/libgat_base.so
  bash scripts/r05_pmc.sh c2 "sq1 sq2 clk sq3 sq4" -- --baseline-config 2
  bash scripts/r05_pmc.sh i8 "sq1 sq2 clk sq3 sq4" -- --layout i8
  cat $OUT/pmc_c2.txt $OUT/pmc_i8.txt
fi
