import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpuacceleratedtracking_amd as g
N, M, L, K, B = 200000, 64, 3, 64, 4
for name, fl in (("full", 0), ("no mfma loop", 1 << 16), ("no fill + no x", (1 << 17) | (1 << 18))):
    op, desc, sig, prm = g.build_stream("GPSL1", N, M, L, K, B, flags=fl, block_seconds=2e-3)
    ctx = op.ctx
    for _ in range(2): op.launch(desc)
    ctx.sync(); ctx.timer_start(); op.launch(desc); ms = ctx.timer_stop()
    info = ctx.last_launch_info()
    nwg = info["workgroups"]
    buf = np.zeros(nwg * 8 * 2, dtype=np.uint64)
    fn = ctx.lib.gat_debug_read; fn.restype = C.c_int32; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert fn(ctx._h, buf.ctypes.data, buf.size) == 0
    d = buf.reshape(nwg, 8, 2).astype(np.float64)
    steps = 782 / info["splits"]
    print(f"{name:16s} {ms:.3f} ms steps/wg {steps:.0f} | per step (memtime ticks @100MHz): consumer work {d[:, :4, 0].mean()/steps:.1f} wait {d[:, :4, 1].mean()/steps:.1f} | producer work {d[:, 4:, 0].mean()/steps:.1f} wait {d[:, 4:, 1].mean()/steps:.1f}")
    del op, desc, sig; torch.cuda.empty_cache()
