#!/usr/bin/env python3
"""Cycle stamps of the split-bf16 kernel's waves (development build -DGAT_MFMA_STAMPS: build/libgat_stamps.so, which runs the
consumers' step loop WITHOUT the cross-barrier pipelining): per role, the share of a step spent working and waiting at the
step barrier, and the producers' work split into carriers + replica / sample split + store.
usage: GAT_LIBRARY=build/libgat_stamps.so python scripts/r05_mfma_stamps.py [planar|i16]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpuacceleratedtracking_amd as g  # noqa: E402

layout_name = sys.argv[1] if len(sys.argv) > 1 else "planar"
layout = {"planar": g.GAT_LAYOUT_PLANAR, "i16": g.GAT_LAYOUT_INTERLEAVED_I16, "i8": g.GAT_LAYOUT_INTERLEAVED_I8}[layout_name]
lib = g.load_library()
op, desc, sig, prm = g.build_stream("GPSL1", 2000000, 64, 3, 64, 1, layout=layout, block_seconds=20e-3)
ctx = op.ctx
for _ in range(20):
    op.launch(desc)
ctx.sync()
info = ctx.last_launch_info()
wgs, waves = info["workgroups"], info["threads"] // 64
buf = np.zeros(wgs * 16 * 4, dtype=np.uint64)
fn = lib.gat_debug_read
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
fn.restype = C.c_int32
rc = fn(ctx._h, buf.ctypes.data_as(C.c_void_p), buf.size)
assert rc == 0, rc
d = buf.reshape(wgs, 16, 4).astype(np.float64)[:, :waves]
steps = -(-2000000 // 32) / info["splits"]
print(f"{layout_name}: {info}, ~{steps:.0f} steps per workgroup")
tot = (d[:, :, 0] + d[:, :, 1])
print("cycles per step and wave (s_memtime ticks: 100 MHz on this part -> x ~23 for shader cycles), mean over workgroups:")
for name, sl in (("consumers (waves 0-3)", slice(0, 4)), ("producers, item waves (4-9)", slice(4, 10)), ("producers, sample waves (10-15)", slice(10, waves))):
    w = d[:, sl]
    print(f"  {name:34s} work {w[:, :, 0].mean() / steps:8.2f} (slowest wave of the role {w[:, :, 0].max(axis=1).mean() / steps:8.2f})  barrier wait {w[:, :, 1].mean() / steps:8.2f}  "
          f"[carriers+replica {w[:, :, 2].mean() / steps:8.2f}  split+store {w[:, :, 3].mean() / steps:8.2f}]  total {tot[:, sl].mean() / steps:8.2f}")
per_wave = d[:, :, 0].mean(axis=0) / steps
print("  work per wave:", " ".join(f"{x:.0f}" for x in per_wave))
