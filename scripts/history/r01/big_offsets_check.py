"""64-bit addressing check: streams far larger than 4 GiB, parity of the LAST blocks against the oracle (the bench and
the tests only look at the first blocks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpuacceleratedtracking_amd as g
import oracle

for (name, N, M, L, K, B, layout) in (("GPSL1", 20000, 4, 3, 1, 16384, 0),      # 10.5 GB planar, vector kernel
                                      ("GPSL1", 50000, 16, 3, 4, 2048, 0),     # 13 GB planar, split-bf16 kernel
                                      ("GPSL1", 50000, 16, 3, 4, 4096, 2)):    # 13 GB int16 pairs, split-bf16 kernel
    op, desc, sig, prm = g.build_stream(name, N, M, L, K, B, layout=layout)
    op.launch(desc)
    got = op.result()
    info = op.ctx.last_launch_info()
    nb = 2
    if layout == 0:
        re = sig[0][:, (B - nb) * N:].cpu().numpy(); im = sig[1][:, (B - nb) * N:].cpu().numpy()
        gb = sig[0].numel() * 8 / 2 ** 30
    else:
        x = sig[0][:, (B - nb) * N:, :].cpu().numpy().astype(np.float32); re, im = x[..., 0].copy(), x[..., 1].copy()
        gb = sig[0].numel() * sig[0].element_size() / 2 ** 30
    p = prm[B - nb:]
    oprm = oracle.make_params(p["prn"], p["code_freq_hz"], p["carrier_freq_hz"], p["code_phase_chips"], p["carrier_phase_cycles"])
    ref = oracle.correlate_f64(re, im, op.system.codes, oprm, N / 1e-3, op.shifts, N=N)
    err = np.max(np.abs(got[B - nb:] - ref) / np.abs(ref).max(axis=(2, 3), keepdims=True))
    print(f"{name} N={N} M={M} K={K} B={B} layout={layout}: {gb:.1f} GiB, kernel {info['matrix_core']}, last-block max rel err {err:.2e}")
    assert err <= 1e-5
    del op, desc, sig; torch.cuda.empty_cache()
print("ok")
