#!/bin/bash
# A/B of the three kernels on the antenna-rich shapes: GAT_MC_MODE 0 = vector, 2 = f32 MFMA, 3 = split-bf16 MFMA
run() { for ev in ${MODES:-GAT_MC_MODE=3 GAT_MC_MODE=2 GAT_MC_MODE=0}; do env $ev timeout 180 python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('[$ev]', c['workload'][:70], '| value %.1f Msamples/s rtf %.1f  %.1f GB/s frac %.3f ms %.4f err %.1e mc=%d wg=%d splits=%d lds=%d' % (d['value'], d['real_time_factor'], r['achieved'], r['frac'], r['kernel_ms_per_launch'], d['parity_max_rel_err_vs_f64_oracle'], c['launch']['matrix_core'], c['launch']['workgroups'], c['launch']['splits'], c['launch']['lds_bytes']))"; done; }
run --gnss GPSL1 --num-samples 50000 --num-ants 16 --num-taps 3 --channels 4  --blocks 512        # C4 per GPU
run --gnss GPSL1 --num-samples 50000 --num-ants 16 --num-taps 3 --channels 32 --blocks 128       # C4 all 32 PRNs on one GPU
run --gnss GPSL1 --num-samples 2000000 --num-ants 64 --num-taps 3 --channels 8 --blocks 2 --block-ms 20   # C5 (8 of 64 channels)
run --gnss GPSL1 --num-samples 2000000 --num-ants 64 --num-taps 3 --channels 64 --blocks 1 --block-ms 20  # C5 full
