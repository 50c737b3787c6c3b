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
SOURCES = ["gat_dc_f0.hip", "gat_dc_f1.hip", "gat_dc_f2.hip", "gat_dc_f3.hip", "gat_resident_f0.hip", "gat_resident_f1.hip",
           "gat_resident_f2.hip", "gat_resident_f3.hip",
           "gat_kernels.hip", "gat_mfma.hip", "gat_mfma_bf16.hip", "gat_api.cpp", "gat_planner.cpp", "gat_group.cpp",
           "gat_resident_api.cpp", "gat_codes.cpp"]
# gat_version.cpp is not in SOURCES: it is compiled at every link with the build's identity (git commit, flags)
HEADERS = [os.path.join(CSRC, "gat_internal.h"), os.path.join(CSRC, "gat_phase.h"), os.path.join(CSRC, "gat_dc.h"),
           os.path.join(CSRC, "gat_dc_body.inc"), os.path.join(CSRC, "gat_resident.h"), os.path.join(CSRC, "gat_ctx.h"),
           os.path.join(ROOT, "include", "gat.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _flags(extra: tuple[str, ...] = ()) -> list[str]:
    # -ffp-contract=off: the double-precision code phase must not be fused (gat_kernels.hip);
    # fused multiply-adds in the hot loops are written explicitly with __builtin_fmaf.
    return ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-DGAT_BUILD",
            "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *extra]


def source_identity() -> str:
    """Short commit of the last change to the kernel sources (csrc/, include/), '+dirty' when the working tree differs from
    it there; 'unknown' outside a git checkout (the GPU box receives the built .so, not .git)."""
    paths = ["gpuacceleratedtracking_amd/csrc", "include", "gpuacceleratedtracking_amd/build.py"]
    try:
        sha = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", *paths], capture_output=True, text=True,
                             check=True).stdout.strip()
        if not sha:
            return "unknown"
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", *paths], capture_output=True, text=True,
                               check=True).stdout.strip()
        return sha + ("+dirty" if dirty else "")
    except Exception:
        return "unknown"


def repo_head() -> str:
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              check=True).stdout.strip() or "unknown"
    except Exception:
        return "unknown"


BUILD_INFO = os.path.join(HERE, "_build_info.json")  # git-ignored, travels with the .so: what built it, from which commit


def build_info() -> dict:
    """The record build_libgat() left next to libgat.so (bench lines and benchmark records quote it)."""
    import json

    try:
        with open(BUILD_INFO) as f:
            return json.load(f)
    except Exception:
        return {}


def command(out: str = LIB) -> list[str]:
    """The one-line recipe (what INTEGRATION.md quotes); build_libgat() runs the same flags per source
    so that an edit to one kernel file does not recompile the others."""
    return [hipcc_path(), *_flags(), "-fno-slp-vectorize", "-shared", *[os.path.join(CSRC, s) for s in SOURCES],
            os.path.join(CSRC, "gat_version.cpp"), "-ldl", "-o", out]


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.join(CSRC, "gat_version.cpp")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_libgat(force: bool = False, verbose: bool = False, extra_flags: tuple[str, ...] = (), out: str = LIB) -> str:
    """Compile libgat.so for gfx950 if missing or older than its sources: one object per source under
    build/obj/ (compiled concurrently, reused while newer than source + headers), then one link.
    ``extra_flags`` / ``out``: diagnostic variants (e.g. -DGAT_MFMA_STAMPS -> build/libgat_stamps.so)."""
    if not (force or is_stale(out)):
        return out
    from concurrent.futures import ThreadPoolExecutor

    if extra_flags and "-DGAT_DEV" not in extra_flags:
        extra_flags = ("-DGAT_DEV",) + tuple(extra_flags)  # every variant is a development build and says so (gat_version)

    tag = "obj" + ("_" + "_".join(f.lstrip("-").replace("=", "_") for f in extra_flags) if extra_flags else "")
    objdir = os.path.join(ROOT, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS + [os.path.abspath(__file__)])
    jobs, objs = [], []
    dc_only = bool(extra_flags) and all(f.startswith("-DGAT_DC_") or f == "-DGAT_DEV" for f in extra_flags)
    # ... and flags that only touch the split-bf16 matrix-core kernel (-DGAT_ABLATE=, -DGAT_MB_*)
    mb_only = bool(extra_flags) and all(f.startswith("-DGAT_ABLATE") or f.startswith("-DGAT_MB_") or f == "-DGAT_DEV" for f in extra_flags)
    base_objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(base_objdir, exist_ok=True)
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        # flags that only touch the fused vector kernel (-DGAT_DC_*): every other object is shared with the main build
        vector_tu = src.startswith("gat_dc_f") or src.startswith("gat_resident_f")  # both are made of gat_dc_body.inc
        shared = dc_only and not vector_tu and src not in ("gat_api.cpp", "gat_planner.cpp", "gat_resident_api.cpp")  # the planner shares gat_internal.h
        shared = shared or (mb_only and src not in ("gat_mfma_bf16.hip", "gat_api.cpp", "gat_planner.cpp"))
        obj = os.path.join(base_objdir if shared else objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if shared:
            extra_here = ()
        else:
            extra_here = extra_flags
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_t):
            # the fused vector kernel is written with scalar FMAs on purpose (gat_dc.h): keep the SLP vectoriser from
            # re-packing them into v_pk_fma_f32 + operand-pairing moves
            per_file = ("-fno-slp-vectorize",) if vector_tu else ()
            jobs.append([hipcc_path(), *_flags(tuple(extra_here) + per_file), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    # build identity (gat_version): compiled at every link
    ident, flag_str = source_identity(), (" ".join(extra_flags) if extra_flags else "none")
    vobj = os.path.join(objdir, "gat_version.o")
    run(["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DGAT_BUILD", "-I" + os.path.join(ROOT, "include"),
         f'-DGAT_GIT_SHA="{ident}"', f'-DGAT_BUILD_FLAGS="{flag_str}"', "-c", os.path.join(CSRC, "gat_version.cpp"), "-o", vobj])
    run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, vobj, "-ldl", "-o", out])
    if out == LIB:
        import json
        import time

        try:
            hipcc_v = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout.splitlines()[0].strip()
        except Exception:
            hipcc_v = "unknown"
        with open(BUILD_INFO, "w") as f:
            json.dump({"kernel_sources_git": ident, "repo_head_git": repo_head(), "flags": flag_str, "hipcc": hipcc_v,
                       "recipe": " ".join(_flags()[:-2]), "built_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}, f)
    return out


C_EXAMPLES = ("gat_known_answer", "gat_multi_gpu", "gat_latency")


def build_c_example(force: bool = False, name: str = "gat_known_answer") -> str:
    """gcc build of examples/<name>.c against libgat.so (plain C hosts of the C ABI: the reference's known answer,
    channels sharded over every GPU of the node, single-block latency from native code)."""
    src = os.path.join(ROOT, "examples", name + ".c")
    out = os.path.join(ROOT, "build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(LIB)):
        subprocess.run(["gcc", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", out, "-L" + HERE,
                        "-lgat", "-Wl,-rpath," + HERE, "-lm"], check=True)
    return out


def build_c_examples(force: bool = False) -> list[str]:
    return [build_c_example(force, n) for n in C_EXAMPLES if os.path.exists(os.path.join(ROOT, "examples", n + ".c"))]


if __name__ == "__main__":
    import sys

    if "--variant" in sys.argv:  # development: python -m ...build --variant NAME -DGAT_DC_DEV -D... -> build/libgat_NAME.so
        i = sys.argv.index("--variant")
        name, flags = sys.argv[i + 1], tuple(a for a in sys.argv[i + 2:] if a.startswith("-"))
        print(build_libgat(extra_flags=flags, out=os.path.join(ROOT, "build", f"libgat_{name}.so"), verbose=False))
        sys.exit(0)

    if "--stamps" in sys.argv:  # diagnostic build of the matrix kernels with per-wave cycle stamps
        print(build_libgat(extra_flags=("-DGAT_MFMA_STAMPS",), out=os.path.join(ROOT, "build", "libgat_stamps.so"),
                           verbose=True))
    else:
        print(build_libgat(force="--force" in sys.argv, verbose=True))
