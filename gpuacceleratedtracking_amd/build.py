"""Build recipe for libgat.so (the C-ABI shared library, include/gat.h).

hipcc cross-compiles for gfx950 without a GPU; the .so is built IN-TREE next to this file so
that it travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgat.so")
SOURCES = ["gat_kernels.hip", "gat_mfma.hip", "gat_api.cpp", "gat_codes.cpp"]
HEADERS = [os.path.join(CSRC, "gat_internal.h"), os.path.join(ROOT, "include", "gat.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def command(out: str = LIB) -> list[str]:
    # -ffp-contract=off: the double-precision code phase must not be fused (gat_kernels.hip);
    # fused multiply-adds in the hot loop are written explicitly with __builtin_fmaf.
    return [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
            "-fvisibility=hidden", "-DGAT_BUILD", "-ffp-contract=off",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
            *[os.path.join(CSRC, s) for s in SOURCES], "-o", out]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_libgat(force: bool = False, verbose: bool = False) -> str:
    """Compile libgat.so for gfx950 if missing or older than its sources."""
    if force or is_stale():
        cmd = command()
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_c_example(force: bool = False) -> str:
    """gcc build of examples/gat_known_answer.c against libgat.so (plain C host of the C ABI)."""
    src = os.path.join(ROOT, "examples", "gat_known_answer.c")
    out = os.path.join(ROOT, "build", "gat_known_answer")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(LIB)):
        subprocess.run(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", out, "-L" + HERE,
                        "-lgat", "-Wl,-rpath," + HERE, "-lm"], check=True)
    return out


if __name__ == "__main__":
    print(build_libgat(force=True, verbose=True))
