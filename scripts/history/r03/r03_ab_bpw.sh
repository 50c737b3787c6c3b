mkdir -p gpurun_out/r03
out=gpurun_out/r03/ab_bpw.txt; : > $out
one() { GAT_DC_BPW_FORCE=$1 timeout -k 10 280 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; l=d['config']['launch']
print('bpw_force %-2s %-14s wgs %-5d bpw %-2d depth %d %.4f ms  %s %.3f err %.1e' % ('$1', '$2' or 'c2', l['workgroups'], l['blocks_per_wg'], l['prefetch_depth'], r['kernel_ms_per_launch'], r['bound'], r['frac'], d['parity_max_rel_err_vs_f64_oracle']))" >> $out; }
for round in 1 2; do for f in 0 2 4; do one $f "--layout i16"; one $f "--layout i8"; one $f ""; done; done
sort -k4,4 -k2,2n -s $out
