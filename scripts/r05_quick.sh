#!/bin/bash
# quick A/B line per shape: scripts/r05_quick.sh TAG [shapes...]  (GAT_LIBRARY selects the build; QARGS: extra bench.py arguments,
# e.g. QARGS="--option dc_aw2=1")
tag=$1; shift
mkdir -p gpurun_out/r05
out=gpurun_out/r05/quick_$tag.txt; : > $out
run() { name=$1; shift; python bench.py --no-cpu-baseline --no-single-block --no-read-ceiling "$@" $QARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; l=d['config']['launch']; s=d['step_ms']
print('%-10s %-5s median %.4f min %.4f mean %.4f ms frac %.4f (%s) hbm_frac %.4f err %.2e  wg %d thr %d tile %d kt %d lds %d  %s %s' % (sys.argv[1], sys.argv[2], s['median'], s['min'], s['mean'], r['frac'], r['bound'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle'], l['workgroups'], l['threads'], l['ant_tile'], l['channels_per_wg'], l['lds_bytes'], d.get('libgat','').split('flags:')[-1], ' '.join(d['config'].get('options', []))))" $tag $name | tee -a $out; }
for s in "$@"; do
case $s in
c1) run c1 ;;
c0) run c0 --num-samples 4000 --num-ants 1 --blocks 16384 ;;
c2) run c2 --baseline-config 2 ;;
c3) run c3 --baseline-config 3 ;;
c3k32) run c3k32 --num-samples 50000 --num-ants 16 --num-taps 3 --channels 32 --blocks 512 ;;     # configs[3] as a whole on one GPU: 16 antennas x 32 PRNs
c2l1) run c2l1 --gnss GPSL1 --num-samples 50000 --num-ants 4 --num-taps 5 --channels 12 --blocks 1024 ;;  # configs[2]'s shape on 1 KB chip tables
c1k8) run c1k8 --channels 8 --blocks 1024 ;;  # configs[1]'s tile with eight channels
c1k2) run c1k2 --channels 2 --blocks 2048 ;;
c1k3) run c1k3 --channels 3 --blocks 2048 ;;
c1k4) run c1k4 --channels 4 --blocks 1024 ;;
c1k5) run c1k5 --channels 5 --blocks 1024 ;;
c1k7) run c1k7 --channels 7 --blocks 1024 ;;
i16k8) run i16k8 --layout i16 --channels 8 --blocks 1024 ;;
ilk8) run ilk8 --layout interleaved --channels 8 --blocks 1024 ;;
m8k4) run m8k4 --num-ants 8 --channels 4 --blocks 1024 ;;
m12k4) run m12k4 --num-ants 12 --channels 4 --blocks 512 ;;
c2i16) run c2i16 --gnss GPSL5 --num-samples 50000 --num-ants 4 --num-taps 5 --channels 12 --blocks 1024 --layout i16 ;;
lat12) run lat12 --channels 12 --blocks 1 --steps 400 ;;   # one 20 MHz block, 12 channels per launch (device time, pipelined)
lat4) run lat4 --channels 4 --blocks 4 --steps 400 ;;
i8k8) run i8k8 --layout i8 --channels 8 --blocks 1024 ;;
c4) run c4 --baseline-config 4 ;;
m32k64) run m32k64 --num-samples 2000000 --num-ants 32 --channels 64 --blocks 1 --block-ms 20 ;;   # one column group fewer rows: 3 groups x 2 row tiles
i8m16k8) run i8m16k8 --num-samples 50000 --num-ants 16 --channels 8 --blocks 64 --layout i8 ;;       # int8 pairs: the matrix kernel by default from 24 columns on
c4i8) run c4i8 --baseline-config 4 --layout i8 ;;
c4i16) run c4i16 --baseline-config 4 --layout i16 ;;
c4k32i16) run c4k32i16 --num-samples 2000000 --num-ants 64 --channels 32 --blocks 1 --block-ms 20 --layout i16 --matrix-core 3 ;;
m32k32i16) run m32k32i16 --num-samples 2000000 --num-ants 32 --channels 32 --blocks 1 --block-ms 20 --layout i16 --matrix-core 3 ;;
i8) run i8 --layout i8 ;;
i16) run i16 --layout i16 ;;
il) run il --layout interleaved ;;
esac
done
