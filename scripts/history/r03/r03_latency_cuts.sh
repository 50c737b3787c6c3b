#!/bin/bash
# Round 3: where the single-block latency goes.  build/gat_latency against the product library and against the four
# diagnostic builds that end the vector kernel early (-DGAT_DC_LAT_CUT=1..4: entry / block set-up + chip table /
# first replica segment + carrier anchors / step loop).  One box, one process at a time.
#   python -m gpuacceleratedtracking_amd.build --variant latcutN -DGAT_DC_LAT_CUT=N   (N = 1..4) first
mkdir -p gpurun_out/r03
out=gpurun_out/r03/latency_cuts.txt
: > $out
reps=${1:-3000}
for v in ${VARIANTS:-main latcut1 latcut2 latcut3 latcut4 latcut5 latcut6 main}; do
  echo "== $v" >> $out
  if [ $v = main ]; then timeout -k 10 200 build/gat_latency $reps >> $out 2>&1 || exit 1
  else # the example finds libgat.so through its RUNPATH; LD_LIBRARY_PATH goes first
    mkdir -p build/variant_$v && cp build/libgat_$v.so build/variant_$v/libgat.so
    LD_LIBRARY_PATH=$PWD/build/variant_$v:$LD_LIBRARY_PATH timeout -k 10 200 build/gat_latency $reps >> $out 2>&1 || exit 1; fi
done
cat $out
