"""In-kernel cycle stamps of the split-bf16 matrix kernel (diagnostic build: python -m
gpuacceleratedtracking_amd.build --stamps; run with GAT_LIBRARY=build/libgat_stamps.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpuacceleratedtracking_amd as g

SHAPES = [  # N, M, L, K, B, block_seconds
    (50000, 16, 3, 4, 512, 1e-3),      # C4 per GPU
    (50000, 16, 3, 32, 128, 1e-3),     # C4 all channels
    (400000, 64, 3, 64, 2, 4e-3),      # C5-like (shorter)
]
for N, M, L, K, B, bs in SHAPES:
    layout = {"planar": 0, "interleaved": 1, "i16": 2, "i8": 3}[os.environ.get("GAT_STAMP_LAYOUT", "planar")]
    op, desc, sig, prm = g.build_stream("GPSL1", N, M, L, K, B, block_seconds=bs, layout=layout)
    ctx = op.ctx
    ctx.set_matrix_core(3)
    for _ in range(2): op.launch(desc)
    ctx.sync(); ctx.timer_start(); op.launch(desc); ms = ctx.timer_stop()
    info = ctx.last_launch_info()
    nwg = info["workgroups"]
    buf = np.zeros(nwg * 16 * 4, dtype=np.uint64)
    fn = ctx.lib.gat_debug_read; fn.restype = C.c_int32; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    assert fn(ctx._h, buf.ctypes.data, buf.size) == 0
    d = buf.reshape(nwg, 16, 4).astype(np.float64)
    rt = info["ant_tile"] // 16
    print(f"N={N} M={M} K={K} B={B}: {ms:.3f} ms, {nwg} wgs, splits {info['splits']}, lds {info['lds_bytes']}, mc {info['matrix_core']}")
    tot = (d[:, :, 0] + d[:, :, 1]).mean()
    print(f"   memtime ticks (100 MHz) per wg: total {tot:.0f} | consumer work {d[:, :4, 0].mean():.0f} wait {d[:, :4, 1].mean():.0f}"
          f" | producer work {d[:, 4:, 0].mean():.0f} (gen {d[:, 4:, 2].mean():.0f}, split/store {d[:, 4:, 3].mean():.0f}) wait {d[:, 4:, 1].mean():.0f}")
    del op, desc, sig; torch.cuda.empty_cache()
