import sys, numpy as np, torch
sys.path.insert(0, '.')
import gpuacceleratedtracking_amd as g, oracle
from tests.helpers import make_case, oracle_result
for if_hz in (0.0, 1e5, 1e6, 4.3e6):
    case = make_case(123, N=20000, M=4, L=3, K=1, B=2, if_hz=if_hz)
    sysobj = g.GPSL1()
    dev = g.get_context().device
    re = torch.from_numpy(case["re"]).to(dev); im = torch.from_numpy(case["im"]).to(dev)
    op = g.StreamCorrelator(sysobj, 20000, 4, 2, 1, case["shifts"], case["fs"])
    op.out_re.fill_(-7.0); op.out_im.fill_(-9.0)
    p = case["prm"]
    op.set_params(g.make_params(p["prn0"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"]))
    op(re, im)
    got = op.result(); ref = oracle_result(case)
    print(if_hz, op.ctx.last_launch_info())
    print(" got", got[0,0,1]); print(" ref", ref[0,0,1])
