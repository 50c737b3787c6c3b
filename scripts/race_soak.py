"""Race soak: the split-bf16 kernel (and the vector kernel) launched many times on the same inputs must give
bit-identical outputs every time (deterministic two-stage sums; any LDS / barrier race between the producer and
consumer waves would show up as a flipped bit sooner or later)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gpuacceleratedtracking_amd as g

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
shapes = [("GPSL1", 200000, 64, 3, 64, 1, 0, 2e-3), ("GPSL1", 200000, 64, 3, 64, 1, 3, 2e-3), ("GPSL1", 50000, 16, 3, 4, 64, 0, 1e-3),
          ("GPSL1", 50000, 16, 3, 32, 16, 2, 1e-3), ("GPSL5", 50000, 4, 5, 12, 64, 0, 1e-3), ("GPSL1", 20000, 4, 3, 1, 512, 0, 1e-3),
          # one-wave workgroups (short blocks, long stream), several blocks per four-wave workgroup, two-antenna tiles, int8
          ("GPSL1", 4000, 1, 3, 1, 16384, 0, 1e-3), ("GPSL1", 4000, 1, 3, 1, 2048, 0, 1e-3), ("GPSL1", 2048, 2, 7, 3, 4096, 3, 1e-3)]
for (name, N, M, L, K, B, layout, bs) in shapes:
    op, desc, sig, prm = g.build_stream(name, N, M, L, K, B, layout=layout, block_seconds=bs)
    op.launch(desc)
    ref_re, ref_im = op.out_re.clone(), op.out_im.clone()
    bad = 0
    for i in range(reps):
        op.launch(desc)
        if not (torch.equal(op.out_re, ref_re) and torch.equal(op.out_im, ref_im)):
            bad += 1
    print(f"{name} N={N} M={M} L={L} K={K} B={B} layout={layout}: kernel {op.ctx.last_launch_info()['matrix_core']}, {reps} launches, {bad} differing")
    assert bad == 0
    del op, desc, sig; torch.cuda.empty_cache()
print("ok")
