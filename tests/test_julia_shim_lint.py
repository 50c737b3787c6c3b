"""Static lint of the Julia shim (julia/GATHip.jl, julia/GATHipHarness.jl) against include/gat.h.

No `julia` binary exists in this project's images, so the shim cannot be executed; what can be checked is checked
here on the CPU: every `ccall((:sym, libgat), Ret, (Types...), args...)` names an exported symbol, passes as many
arguments as the C prototype has, with matching type classes (Ptr/Ref/Cstring <-> pointer, Int32 <-> int32_t,
Int64 <-> int64_t, Float64 <-> double, Csize_t <-> size_t ...); the `struct` mirrors have the header's fields in
the header's order with C layout (sizes / offsets from gcc); the constants agree; every export is bound; and the
harness defines the methods the reference's run_kernel_benchmark calls (/root/reference is NOT read: the three names
are fixed by src/benchmarks.jl:963-979 and quoted in julia/GATHipHarness.jl).
"""
from __future__ import annotations

import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gat.h")
JULIA = [os.path.join(ROOT, "julia", f) for f in ("GATHip.jl", "GATHipHarness.jl")]


# ---------------------------------------------------------------------------------------------- C side
def _strip_c_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_class(ctype: str) -> str:
    t = " ".join(ctype.replace("const", " ").split())
    if "*" in t:
        return "ptr"
    return {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "size", "double": "f64",
            "float": "f32", "void": "void"}[t]


def c_prototypes() -> dict[str, tuple[str, list[str]]]:
    """name -> (return class, [argument classes]) of every GAT_API function."""
    text = _strip_c_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"GAT_API\s+([\w\s\*]+?)\b(gat_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        arg_classes = []
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name: last identifier, unless the declarator is a pure type
                mm = re.match(r"(.*?)(\w+)$", a)
                ctype = mm.group(1).strip() if mm and mm.group(1).strip() else a
                arg_classes.append(c_class(ctype))
        protos[name] = (c_class(ret), arg_classes)
    return protos


def c_structs() -> dict[str, list[tuple[str, str]]]:
    """struct name -> [(field C type, field name)] in declaration order."""
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(gat_\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "int32_t num_taps, early_index, ..." or "const void *re"
            if "*" in decl:
                t, n = decl.rsplit("*", 1)
                fields.append((t.strip() + " *", n.strip()))
            else:
                t, names = decl.split(" ", 1)
                for n in names.split(","):
                    fields.append((t, n.strip()))
        out[m.group(1)] = fields
    return out


def c_defines() -> dict[str, int]:
    text = _strip_c_comments(open(HEADER).read())
    out = {}
    for m in re.finditer(r"#define\s+(GAT_\w+)\s+(\d+)u?\s*$", text, flags=re.M):
        out[m.group(1)] = int(m.group(2))
    return out


def c_layout(struct: str, fields: list[str]) -> tuple[int, list[int]]:
    """sizeof and offsetof of every field, from gcc."""
    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "gat.h"', "int main(void){",
            f'printf("%zu\\n", sizeof({struct}));']
    prog += [f'printf("%zu\\n", offsetof({struct}, {f}));' for f in fields]
    prog += ["return 0;}"]
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "l.c"), os.path.join(td, "l")
        open(src, "w").write("\n".join(prog))
        subprocess.run(["gcc", "-I" + os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        nums = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    return nums[0], nums[1:]


# ------------------------------------------------------------------------------------------ Julia side
def _strip_jl_comments(text: str) -> str:
    return "\n".join(line.split("#", 1)[0] if '"' not in line.split("#", 1)[0] or line.split("#", 1)[0].count('"') % 2 == 0
                     else line for line in text.splitlines())


def _split_top(s: str) -> list[str]:
    """split at top-level commas (parentheses, braces and brackets nest)"""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        parts.append("".join(cur).strip())
    return parts


def jl_class(t: str) -> str:
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return {"Int32": "i32", "Cint": "i32", "UInt32": "u32", "Int64": "i64", "UInt64": "u64", "Csize_t": "size",
            "Float64": "f64", "Cdouble": "f64", "Cfloat": "f32", "Float32": "f32", "Cvoid": "void", "Int8": "i8", "UInt8": "u8"}[t]


def jl_ccalls() -> list[dict]:
    calls = []
    for path in JULIA:
        text = _strip_jl_comments(open(path).read())
        for m in re.finditer(r"ccall\(", text):
            i, depth = m.end(), 1
            while depth:  # matching parenthesis of the ccall
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            parts = _split_top(text[m.end():i - 1])
            mm = re.match(r"\(\s*:(\w+)\s*,\s*libgat\s*\)", parts[0])
            assert mm, f"{os.path.basename(path)}: ccall target is not (:symbol, libgat): {parts[0]!r}"
            types = parts[2].strip()
            assert types.startswith("(") and types.endswith(")"), parts[2]
            argtypes = _split_top(types[1:-1])
            line = text.count("\n", 0, m.start()) + 1
            calls.append(dict(file=os.path.basename(path), line=line, sym=mm.group(1), ret=parts[1].strip(),
                              argtypes=argtypes, nargs=len(parts) - 3))
    return calls


def jl_structs() -> dict[str, list[tuple[str, str]]]:
    text = _strip_jl_comments(open(JULIA[0]).read())
    out = {}
    for m in re.finditer(r"^struct\s+(\w+)\s*\n(.*?)^end", text, flags=re.S | re.M):
        fields = [(f.group(2), f.group(1)) for f in re.finditer(r"^\s*(\w+)::([\w\{\}]+)", m.group(2), flags=re.M)]
        out[m.group(1)] = fields
    return out


JL_SIZE = {"Int32": 4, "UInt32": 4, "Int64": 8, "UInt64": 8, "Float64": 8, "Float32": 4, "Cfloat": 4, "Ptr{Cvoid}": 8}


def jl_layout(fields: list[tuple[str, str]]) -> tuple[int, list[int]]:
    """C-compatible layout of an isbits Julia struct: natural alignment, tail padded to the largest member."""
    off, offs, amax = 0, [], 1
    for t, _ in fields:
        sz = JL_SIZE[t]
        off = (off + sz - 1) // sz * sz
        offs.append(off)
        off += sz
        amax = max(amax, sz)
    return (off + amax - 1) // amax * amax, offs


STRUCT_PAIRS = {"gat_channel_params": "ChannelParams", "gat_signal_desc": "SignalDesc", "gat_loop_config": "LoopConfig",
                "gat_loop_state": "LoopState", "gat_launch_info": "LaunchInfo", "gat_resident_config": "ResidentConfig",
                "gat_resident_info": "ResidentInfo"}


# ------------------------------------------------------------------------------------------------ tests
def test_header_parses():
    protos = c_prototypes()
    assert len(protos) >= 44 and "gat_downconvert_and_correlate" in protos and "gat_group_gather" in protos
    assert protos["gat_version"] == ("ptr", [])
    assert protos["gat_downconvert_and_correlate"][1] == ["ptr", "ptr", "ptr", "i32", "i32", "i32", "ptr", "f64", "ptr", "ptr", "u32"]
    assert protos["gat_memcpy_h2d"][1] == ["ptr", "ptr", "ptr", "size"]


def test_every_ccall_matches_its_prototype():
    protos, calls = c_prototypes(), jl_ccalls()
    assert len(calls) >= 45
    for c in calls:
        where = f"{c['file']}:{c['line']} ccall :{c['sym']}"
        assert c["sym"] in protos, f"{where}: not exported by include/gat.h"
        ret, args = protos[c["sym"]]
        assert len(c["argtypes"]) == len(args), f"{where}: {len(c['argtypes'])} argument types, the prototype has {len(args)}"
        assert c["nargs"] == len(args), f"{where}: {c['nargs']} argument values for {len(args)} parameters"
        assert jl_class(c["ret"]) == ret, f"{where}: return {c['ret']} vs {ret}"
        for i, (jt, ct) in enumerate(zip(c["argtypes"], args)):
            assert jl_class(jt) == ct, f"{where}: argument {i + 1} is {jt} ({jl_class(jt)}), the header says {ct}"


def test_every_export_is_bound():
    bound = {c["sym"] for c in jl_ccalls()}
    missing = sorted(set(c_prototypes()) - bound)
    assert not missing, f"exports of include/gat.h without a Julia binding: {missing}"


def test_struct_mirrors_have_c_layout():
    cs, js = c_structs(), jl_structs()
    for cname, jname in STRUCT_PAIRS.items():
        assert cname in cs and jname in js, (cname, jname)
        cf, jf = cs[cname], js[jname]
        assert [n for _, n in cf] == [n for _, n in jf], f"{jname}: field names / order differ from {cname}"
        for (ct, n), (jt, _) in zip(cf, jf):
            assert c_class(ct) == jl_class(jt), f"{jname}.{n}: {jt} vs {ct}"
        size, offs = c_layout(cname, [n for _, n in cf])
        jsize, joffs = jl_layout(jf)
        assert (size, offs) == (jsize, joffs), f"{jname}: layout {jsize} {joffs} vs C {size} {offs}"


def test_constants_agree():
    defs = c_defines()
    text = open(JULIA[0]).read()
    for name in ("GAT_OK", "GAT_FLAG_ATOMIC", "GAT_FLAG_GRAPH", "GAT_LAYOUT_PLANAR", "GAT_LAYOUT_INTERLEAVED",
                 "GAT_LAYOUT_INTERLEAVED_I16", "GAT_LAYOUT_INTERLEAVED_I8"):
        m = re.search(rf"const {name} = U?Int32\((\d+)\)", text)
        assert m and int(m.group(1)) == defs[name], name
    m = re.search(r"const GAT_MC_VECTOR, GAT_MC_AUTO, GAT_MC_F32, GAT_MC_BF16_SPLIT = (.*)", text)
    assert [int(x) for x in re.findall(r"Int32\((\d+)\)", m.group(1))] == [defs[k] for k in
                                                                          ("GAT_MC_VECTOR", "GAT_MC_AUTO", "GAT_MC_F32", "GAT_MC_BF16_SPLIT")]


def test_harness_defines_what_run_kernel_benchmark_calls():
    """run_kernel_benchmark (reference src/benchmarks.jl:963-979) calls _run_kernel_benchmark, add_results! (generic) and
    add_metadata!; the last one's reference method needs CUDA.jl, so the harness must bring its own for id 9000 -- filled
    from gat_device_info, with the reference's keys."""
    text = _strip_jl_comments(open(JULIA[1]).read())
    for sig in (r"function _run_kernel_benchmark\(", r"function add_metadata!\(benchmark_results_w_params, processor, algorithm::KernelAlgorithm\{9000\}\)",
                r"function kernel_algorithm\("):
        assert re.search(sig, text), sig
    assert len(re.findall(r"::KernelAlgorithm\{9000\}", text)) == 3
    body = text[text.index("function add_metadata!"):]
    body = body[:body.index("\nend")]
    for key in ('"os"', '"CPU_model"', '"GPU_model"', '"CUDA"', '"HIP"', '"libgat"', '"algorithm"'):
        assert key in body, key
    assert "GATHip.device_info" in body and "GATHip.version()" in body and "CUDA." not in body


@pytest.mark.parametrize("path", JULIA)
def test_julia_blocks_balance(path):
    """cheap syntax guard: block openers and `end`s balance, parentheses balance"""
    text = _strip_jl_comments(open(path).read())
    text = re.sub(r'"(?:[^"\\]|\\.)*"', '""', text)
    assert text.count("(") == text.count(")") and text.count("[") == text.count("]") and text.count("{") == text.count("}")
    # comprehensions (`[f(k) for k in 1:K]`) and `x[end]` live inside brackets: drop bracketed text before counting
    flat, depth = [], 0
    for ch in text:
        depth += ch == "["
        if depth == 0:
            flat.append(ch)
        depth -= ch == "]"
    text = "".join(flat)
    openers = len(re.findall(r"(?<![\w!.])(?:function|struct|if|for|while|let|begin|do|module|try|quote)(?![\w!])", text))
    ends = len(re.findall(r"(?<![\w!.:])end(?![\w!])", text))
    assert openers == ends, (openers, ends)
