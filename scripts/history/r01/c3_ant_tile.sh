for mt in 4 2 1; do GAT_MAX_ANT_TILE=$mt timeout 180 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --gnss GPSL5 --num-samples 50000 --num-ants 4 --num-taps 5 --channels 12 --blocks 1024 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']
print('MT=$mt rtf %.1f ms %.4f err %.1e' % (d['real_time_factor'], r['kernel_ms_per_launch'], d['parity_max_rel_err_vs_f64_oracle']), c['launch'])"; done
