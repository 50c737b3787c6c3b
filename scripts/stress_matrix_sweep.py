"""One-off stress: the random antenna-rich sweep of tests/test_random_sweep_gpu.py over many more seeds (channels whose
single tap cancels to < 1 % of the block's largest accumulator are scaled by that floor: tests/helpers.check_close)."""
import sys, os, time
os.environ.setdefault("GAT_CHECK_FLOOR_FRAC", "0.01")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuacceleratedtracking_amd as g
from tests import test_random_sweep_gpu as t

g.load_library()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
fn = t.test_random_case if len(sys.argv) > 3 and sys.argv[3] == "vector" else t.test_random_matrix_case
bad = []
t0 = time.time()
for seed in range(lo, hi):
    try:
        fn(g, seed)
    except Exception as e:  # noqa: BLE001
        bad.append((seed, str(e)[:300]))
        print("FAIL", seed, str(e)[:300], flush=True)
    if (seed - lo) % 50 == 49:
        print(f"{seed - lo + 1} cases, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
print("done:", hi - lo, "cases,", len(bad), "failures")
sys.exit(1 if bad else 0)
