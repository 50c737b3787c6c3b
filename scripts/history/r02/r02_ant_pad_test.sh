for pad in 0 64 1024 4160 65600; do
  GAT_MC_MODE=0 timeout -k 10 240 python bench.py --no-cpu-baseline --baseline-config 3 --ant-pad $pad 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('c4 pad $pad  %.4f ms hbm %.3f err %.1e' % (r['kernel_ms_per_launch'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))"
done
for pad in 0 1024 4160; do
  timeout -k 10 240 python bench.py --no-cpu-baseline --steps 100 --warmup 20 --ant-pad $pad 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('c2 pad $pad  %.4f ms hbm %.3f err %.1e' % (r['kernel_ms_per_launch'], r['hbm_frac'], d['parity_max_rel_err_vs_f64_oracle']))"
done
