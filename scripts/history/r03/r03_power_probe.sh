#!/bin/bash
# Round 3: is the configs[3] shard power-limited?  Samples rocm-smi (socket power, sclk, mclk) every 0.2 s while
# bench.py loops over one shape; prints the distribution.  usage: r03_power_probe.sh tag bench-args...
tag=$1; shift
mkdir -p gpurun_out/r03
log=gpurun_out/r03/power_$tag.txt
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.2; done ) > $log.raw &
SMI=$!
timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/power_$tag.json 2>/dev/null
kill $SMI
python - $log.raw $tag gpurun_out/r03/power_$tag.json <<'PY' | tee $log
import json, sys, re
rows=[]
for line in open(sys.argv[1]):
    try: d=json.loads(line)["card0"]
    except Exception: continue
    p=[float(v) for k,v in d.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))]
    sclk=[v for k,v in d.items() if k.startswith("sclk")]
    mclk=[v for k,v in d.items() if k.startswith("mclk")]
    rows.append((p[0] if p else -1, sclk[0] if sclk else "", mclk[0] if mclk else ""))
b=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[2], "bench", b["roofline"]["kernel_ms_per_launch"], "ms/launch", b["roofline"]["bound"], b["roofline"]["frac"])
print("samples", len(rows))
ps=sorted(r[0] for r in rows)
if ps: print("power W: min %.0f median %.0f p90 %.0f max %.0f" % (ps[0], ps[len(ps)//2], ps[int(len(ps)*0.9)], ps[-1]))
from collections import Counter
print("sclk", Counter(r[1] for r in rows).most_common(6))
print("mclk", Counter(r[2] for r in rows).most_common(3))
PY
rm -f $log.raw
