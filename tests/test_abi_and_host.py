"""CPU tests: the C-ABI library loads and exports every symbol include/gat.h declares; host-only
entry points (code generators, sample shifts) agree with the oracle; host logic (selectors,
argument forms, sharding plan); and the product fails loudly without a GPU (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    import gpuacceleratedtracking_amd as g
    return g


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "gat.h")).read()
    return sorted(set(re.findall(r"GAT_API\s+[\w\s\*]+?\b(gat_\w+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(g):
    from gpuacceleratedtracking_amd import _lib
    lib = g.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gat.h but not exported by libgat.so"
    assert sorted(_lib.EXPORTS) == syms, "ctypes binding and header disagree"
    assert lib.gat_version().startswith(b"libgat")


def test_in_tree_library_is_a_product_build(g):
    """gat_version() names the library's version, the commit of its kernel sources and every -D flag beyond the product
    recipe: the in-tree libgat.so (what the GPU tests and bench.py load) is built without development flags -- no
    environment knobs (-DGAT_DEV), no diagnostic kernels -- and reads no kernel-selecting environment variable."""
    import re
    import subprocess
    from gpuacceleratedtracking_amd import _lib, build
    v = g.load_library().gat_version().decode()
    m = re.fullmatch(r"libgat (\d+\.\d+\.\d+) \(gfx950\) git:([0-9a-f]{7,12}(?:\+dirty)?|unknown) flags:(.+)", v)
    assert m, v
    assert m.group(3) == "none", f"in-tree libgat.so carries development flags: {v}"
    assert tuple(int(x) for x in m.group(1).split(".")) >= (0, 2, 0)
    # the only environment variable the product library looks at is the tracing toggle
    strings = subprocess.run(["strings", "-a", _lib.library_path()], capture_output=True, text=True, check=True).stdout
    envs = set(re.findall(r"\bGAT_[A-Z0-9_]{3,}\b", strings)) - {"GAT_ROCTX"}
    assert not {e for e in envs if e.startswith(("GAT_DC_", "GAT_MC_MODE", "GAT_NO_MFMA", "GAT_MAX_ANT", "GAT_SYNC_FLAG"))}, envs
    info = build.build_info()
    assert info.get("flags") == "none" and info.get("kernel_sources_git") == m.group(2)
    # development variants say what they are
    assert "-DGAT_DEV" in open(build.__file__).read()


def test_constants_match_header(g):
    """Every numeric #define of include/gat.h that the Python host layer mirrors has the same value there."""
    import re
    from gpuacceleratedtracking_amd import _lib
    text = open(os.path.join(ROOT, "include", "gat.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define\s+(GAT_[A-Z0-9_]+)\s+(-?\d+)u?\b", text, re.M)}
    mirrored = [n for n in defs if hasattr(_lib, n)]
    for name in ("GAT_FLAG_ATOMIC", "GAT_FLAG_GRAPH", "GAT_LAYOUT_PLANAR", "GAT_LAYOUT_INTERLEAVED", "GAT_LAYOUT_INTERLEAVED_I16",
                 "GAT_LAYOUT_INTERLEAVED_I8", "GAT_MC_VECTOR", "GAT_MC_AUTO", "GAT_MC_F32", "GAT_MC_BF16_SPLIT", "GAT_MAX_TAPS"):
        assert name in mirrored, name
    for name in mirrored:
        assert getattr(_lib, name) == defs[name], (name, getattr(_lib, name), defs[name])


def test_struct_layouts_match_header(g):
    from gpuacceleratedtracking_amd import _lib
    assert C.sizeof(_lib.ChannelParams) == 40 and _lib.PARAMS_DTYPE.itemsize == 40
    assert C.sizeof(_lib.SignalDesc) == 56
    assert C.sizeof(_lib.LaunchInfo) == 48
    # the oracle's params struct has the same layout (prn0/pad == prn/reserved)
    assert oracle.PARAMS_DTYPE.itemsize == 40


@pytest.mark.parametrize("system", ["GPSL1", "GPSL5"])
def test_host_code_generators_match_oracle(g, system):
    """libgat's generator (G2-delay / XB-advance formulation) vs the oracle's (tap-selector /
    stepping formulation): two independent implementations, all 37 PRNs."""
    tbl, fc = g.generate_codes(system, 37)
    lc = oracle.SYSTEMS[system][0]
    assert tbl.shape == (37, lc) and fc == oracle.SYSTEMS[system][1]
    lib = oracle.lib()
    fn = getattr(lib, oracle.SYSTEMS[system][2])
    for p in range(37):
        row = np.empty(lc, dtype=np.int8)
        assert fn(p + 1, row.ctypes.data_as(C.POINTER(C.c_int8))) == 0
        assert np.array_equal(row, tbl[p]), f"{system} PRN {p + 1}"


def test_host_l5_generator_against_icd_initial_states(g):
    """libgat's own GPS L5 I5 generator against IS-GPS-705's initial XB code states (tests/golden, PRN 1-37): the
    first 13 code chips are the complement of the state read from stage 13 down (XA = all ones); and its C/A
    generator against IS-GPS-200's first-10-chip octals -- the product's tables are pinned to the ICDs directly, not
    only through the oracle."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
    tbl, _ = g.generate_codes("GPSL5", 37)
    for prn, state in enumerate(gold["l5i_xb_initial_state"], 1):
        want = [1 - 2 * (1 ^ int(c)) for c in reversed(state)]
        assert tbl[prn - 1, :13].tolist() == want, f"PRN {prn}"
    ca, _ = g.generate_codes("GPSL1", 32)
    for p, want in enumerate(gold["ca_first10_octal"]):
        v = 0
        for b in (1 - ca[p, :10].astype(int)) // 2:
            v = (v << 1) | int(b)
        assert oct(v)[2:] == want, f"PRN {p + 1}"


def test_gen_codes_errors(g):
    lib = g.load_library()
    lc, fc = C.c_int32(), C.c_double()
    assert lib.gat_gen_codes(b"GALILEO", 1, None, C.byref(lc), C.byref(fc)) == 4  # GAT_ERR_UNSUPPORTED
    buf = (C.c_int8 * 1023)()
    assert lib.gat_gen_codes(b"GPSL1", 38, buf, C.byref(lc), C.byref(fc)) == 2   # GAT_ERR_RANGE
    assert lib.gat_gen_codes(None, 1, buf, C.byref(lc), C.byref(fc)) == 1         # GAT_ERR_ARG


@pytest.mark.parametrize("L,fs,fc", [(3, 2.5e6, 1.023e6), (3, 20e6, 1.023e6), (5, 50e6, 10.23e6), (7, 16.384e6, 1.023e6),
                                     (1, 4e6, 1.023e6), (4, 4e6, 1.023e6), (3, 1.5345e6, 1.023e6)])
def test_sample_shifts_match_oracle(g, L, fs, fc):
    class Sys:  # duck-typed system
        code_frequency = fc
    corr = g.EarlyPromptLateCorrelator(g.NumAnts(1), g.NumAccumulators(L))
    got = g.get_correlator_sample_shifts(Sys(), corr, fs, 0.5)
    assert got.tolist() == oracle.sample_shifts(L, fs, fc).tolist()


def test_host_loop_update_matches_oracle_restatement(g):
    """gat_tracking_update_host (the device kernel's arithmetic on the host, csrc/gat_loop.h; needs no GPU) against the oracle's
    restatement of the loop equations, several steps so that the filter integrators are exercised; error codes."""
    lib = g.load_library()
    L_ = g._lib
    K, M, L = 5, 3, 3
    rng = np.random.default_rng(4)
    cfg = L_.LoopConfig(1e-3, 18.0, 2.0, 1.023e6, 1575.42e6, 1.0e5, 1.0, 1023, L, 0, 1, 2)
    cfgd = {n: getattr(cfg, n) for n, _ in cfg._fields_}
    dop = rng.uniform(-3e3, 3e3, K)
    cur = g.make_params(np.arange(K), 1.023e6 + dop * 1.023e6 / 1575.42e6, 1.0e5 + dop, rng.uniform(0, 1023, K), rng.uniform(0, 1, K), shape=(K,))
    nxt = cur.copy()
    st = np.zeros(K, dtype=L_.LOOP_STATE_DTYPE)
    st["init_carrier_doppler_hz"] = dop
    st["carrier_doppler_hz"] = dop
    ostate = {n: st[n].copy() for n in st.dtype.names}
    ocur = oracle.make_params(cur["prn"], cur["code_freq_hz"], cur["carrier_freq_hz"], cur["code_phase_chips"], cur["carrier_phase_cycles"])
    vp = C.c_void_p
    for it in range(5):
        acc = (rng.standard_normal((K, L, M)) + 1j * rng.standard_normal((K, L, M))).astype(np.complex64) * 1000
        re, im = np.ascontiguousarray(acc.real), np.ascontiguousarray(acc.imag)
        assert lib.gat_tracking_update_host(vp(re.ctypes.data), vp(im.ctypes.data), K, M, C.byref(cfg), vp(st.ctypes.data), vp(cur.ctypes.data),
                                            vp(nxt.ctypes.data)) == 0
        cur, nxt = nxt, cur
        ocur, ostate = oracle.np_tracking_update(acc, cfgd, ostate, ocur)
        for f in ("code_freq_hz", "carrier_freq_hz", "code_phase_chips", "carrier_phase_cycles"):
            assert np.allclose(cur[f], ocur[f], rtol=1e-12, atol=1e-9), (it, f)
        for name in ostate:
            assert np.allclose(st[name], ostate[name], rtol=1e-10, atol=1e-9), (it, name)
    bad = L_.LoopConfig(1e-3, 18.0, 2.0, 1.023e6, 1575.42e6, 0.0, 1.0, 1023, L, 0, 1, 7)  # late tap outside the list
    assert lib.gat_tracking_update_host(vp(re.ctypes.data), vp(im.ctypes.data), K, M, C.byref(bad), vp(st.ctypes.data), vp(cur.ctypes.data), vp(nxt.ctypes.data)) == 2
    assert lib.gat_tracking_update_host(None, vp(im.ctypes.data), K, M, C.byref(cfg), vp(st.ctypes.data), vp(cur.ctypes.data), vp(nxt.ctypes.data)) == 1


def test_selectors_and_dicts(g):
    assert g.KernelAlgorithm(1330) == g.KernelAlgorithm(1330) != g.KernelAlgorithm(1331)
    assert g.ALGODICT["4_4_cplx_multi_textmem"] == 4431 and g.ALGODICTINV[4431] == "4_4_cplx_multi_textmem"
    assert g.ALGODICTINV[1331] == "1_3_cplx_multi_textmem"  # Julia Dict literal: the later duplicate wins
    assert set(g.GNSSDICT) == {"GPSL1", "GPSL5"}
    assert g.REDDICT["cplx_multi"] == g.ReductionAlgorithm(3) and g.MEMDICT["textmem"] == g.ReplicaAlgorithm(2)
    s = g.GPSL1(use_gpu=False)
    assert g.get_code_length(s) == 1023 and g.get_code_frequency(s) == 1.023e6
    assert g.get_code_length(g.GPSL5()) == 10230


def test_kernel_algorithm_argument_forms(g):
    with pytest.raises(TypeError):
        g.kernel_algorithm(*([None] * 24), g.KernelAlgorithm(4431))  # 4431 takes 26
    with pytest.raises(TypeError):
        g.kernel_algorithm(*([None] * 25), g.KernelAlgorithm(1330))  # 1330 takes 25
    with pytest.raises(NotImplementedError):
        g.kernel_algorithm(*([None] * 25), g.KernelAlgorithm(1300))  # exported name without a method
    with pytest.raises(TypeError):
        g.kernel_algorithm(1, 2, 3)


def test_make_params_and_correlator(g):
    p = g.make_params(np.arange(3), 1.023e6, [1.0, 2.0, 3.0], 0.5, 0.25, shape=(2, 3))
    assert p.shape == (2, 3) and p["prn"].tolist() == [[0, 1, 2]] * 2 and p["carrier_freq_hz"][1].tolist() == [1, 2, 3]
    assert p.dtype.itemsize == 40 and (p["reserved"] == 0).all()
    c = g.EarlyPromptLateCorrelator(g.NumAnts(4), g.NumAccumulators(3))
    assert g.get_num_ants(c) == 4 and g.get_num_accumulators(c) == 3
    assert g.get_accumulators(c).shape == (3, 4) and not g.get_accumulators(c).any()
    with pytest.raises(ValueError):
        g.EarlyPromptLateCorrelator(0, 3)


def test_shard_plan(g):
    for total in (0, 1, 4, 7, 32):
        for world in (1, 2, 3, 8):
            spans = [g.shard_channels(total, world, r).bounds() for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == g.shard_channels(total, world, 0).counts()
    plan = g.shard_channels(32, 8, 3)
    assert (plan.lo, plan.hi, plan.count) == (12, 16, 4)  # BASELINE config 4: 4 PRN/GPU
    prm = g.make_params(np.arange(32), 1.0, 2.0, 3.0, 4.0, shape=(5, 32))
    assert g.shard_params(prm, plan)["prn"][0].tolist() == [12, 13, 14, 15]
    with pytest.raises(ValueError):
        g.ShardPlan(4, 2, 2)


def test_algorithmic_bytes(g):
    # C2 block = 8*20000*4 + 8*4*3*1 = 640 096 B
    assert g.algorithmic_bytes(1, 20000, 4, 3, 1) == 640096
    assert g.algorithmic_bytes(4096, 20000, 4, 3, 1) == 4096 * 640096


def test_no_gpu_means_loud_failure_not_fallback(g):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.get_context()
    with pytest.raises(RuntimeError):
        g.gen_signal(g.GPSL1(), 1, 1500.0, 2500)
    # and the C ABI itself reports a HIP error instead of inventing a context
    lib = g.load_library()
    h = C.c_void_p()
    assert lib.gat_create(0, None, C.byref(h)) != 0 and not h.value


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gpuacceleratedtracking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "libgat_oracle", "gat_oracle_", "oracle/"):
                    assert needle not in src, f"{f} references the test oracle ({needle})"


def test_c_example_builds_against_the_header():
    """The plain-C host compiles and links against include/gat.h + libgat.so (run on the GPU in
    tests/test_parity_gpu.py::test_c_abi_from_plain_c)."""
    from gpuacceleratedtracking_amd import build
    exes = build.build_c_examples(force=True)
    assert len(exes) >= 2 and all(os.path.exists(e) for e in exes)
