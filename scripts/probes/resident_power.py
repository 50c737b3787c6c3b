import os, sys, time, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpuacceleratedtracking_amd as g
from tests.helpers import make_case
def power():
    try:
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        vals = [l for l in out.splitlines() if "Power" in l and "W" in l]
        return "; ".join(v.strip() for v in vals)[:200]
    except Exception as e:
        return f"rocm-smi failed: {e}"
ctx = g.get_context()
print("idle, no resident kernel:", power()); time.sleep(1.0); print("idle, no resident kernel:", power())
for N, M, wg in ((2048, 4, 0), (262144, 4, 0)):
    case = make_case(1, N=N, M=M, L=3, K=1, B=1)
    ctx.set_codes(case["codes"])
    re = torch.from_numpy(case["re"]).to(ctx.device); im = torch.from_numpy(case["im"]).to(ctx.device)
    torch.cuda.current_stream().synchronize()
    desc = g._lib.SignalDesc(re.data_ptr(), im.data_ptr(), g.GAT_LAYOUT_PLANAR, M, N, N, N, 0)
    p = case["prm"][0]
    prm = g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    with ctx.open_resident(desc, 1, case["shifts"], case["fs"], idle_us=20000000, life_ms=60000) as res:
        res.correlate(prm)
        time.sleep(1.5)
        print(f"resident kernel waiting, {res.info()['workgroups']} workgroups:", power())
        time.sleep(1.0)
        print(f"resident kernel waiting, {res.info()['workgroups']} workgroups:", power())
        t0 = time.time(); n = 0
        while time.time() - t0 < 2.0:
            res.correlate(prm); n += 1
            time.sleep(0.001)
        print(f"  ... serving {n/2:.0f} calls/s:", power())
time.sleep(1.0); print("idle again:", power())
