#!/usr/bin/env python3
"""bench.py -- headline measurement of the downconvert + correlate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works two ways: under an external launcher (python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ..., the driver's form: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or
called plainly as above -- then this process starts N child processes (one rank per GPU) BEFORE anything
touches the GPU, waits for them and exits with their status; it never re-executes itself.

Workload (BASELINE.json configs[1], the one the metric is quoted on): GPS L1 C/A, 4 antennas,
1 PRN per GPU, 3 E/P/L correlators, 1 ms integration blocks at fs = 20 MHz (N = 20 000), as a
batched stream of B consecutive blocks resident in HBM (B = 4096 -> 2.6 GB, far beyond the 256 MB
Infinity Cache).  One "step" = one pass of the fused kernel over all B blocks.

Prints ONE JSON line on rank 0:
  value      = total samples x channels correlated per second over all ranks [Msamples/s]
               (inputs resident in HBM when the timed region starts; sync-inclusive)
  step_ms    = {min, median, mean, sigma, max, n}: every timed step's duration, one HIP event per step on the launch
               stream (the reference keeps every sample's time: BenchmarkTools Minimum / Median / Mean / sigma / Maximum,
               src/benchmarks.jl:1-9)
  roofline   = max(algorithmic bytes / 8 TB/s, algorithmic flops / 157.3 TFLOP/s) against the MEDIAN step duration
               (frac_mean: against the mean of the same steps); `bound` names the winning term; read_ceiling_GBps = what
               a kernel that only reads reaches over the same stream in this process (frac_of_ceiling); terms_ms.valu_issue
               = the kernel's vector instructions (stored PMC count) at one wave-instruction per 2 cycles and SIMD
  cpu_baseline = the oracle's FP32 4-pass CPU restatement ("port") timed on this host on a bounded
               sample of the same stream (the reference's Julia CPU path cannot run here)
  shard_config3 (N > 1 only) = the same measurement, same settle / warm-up / steps, on BASELINE configs[3]'s per-GPU
               shard (16 antennas, 4 of the 32 PRNs per GPU, 1 ms @ 50 MHz)
  constellation_config3 (--constellation, and by default for N > 1) = STRONG scaling of BASELINE configs[3] as a whole:
               32 PRNs x 16 antennas @ 50 MHz, ShardPlan(32, N, rank) -> 32 / 16 / 8 / 4 PRNs per GPU at N = 1 / 2 / 4 / 8,
               same protocol; value and real-time factor are the whole receiver's
  group_check (N > 1 only) = build/gat_multi_gpu run in a FRESH process after the ranks have finished: every visible
               device as one device group from one host thread -- peer replication of the signal (hipMemcpyPeerAsync
               between distinct devices), sharded launch, gather; bit_identical vs shard-by-shard on device 0, peer_copy_GBps
  ranks (N > 1 only) = per rank: ms per step, device name, PCI bus id; world size and backend
Multi-GPU: satellite channels shard with no collective; every rank holds the full antenna signal
and correlates its own PRNs ("weak" scaling: per-GPU work fixed).  RCCL carries only the timing
barrier and the max-over-ranks reduction.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
F32_PEAK_TFLOPS = 157.3   # vector FP32 == f32-input MFMA peak (spec)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (spec)
NUM_SIMDS, PEAK_CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2.0  # 256 CUs x 4 SIMD-32; a wave64 vector instruction issues over 2 cycles

CONSTELLATION_PRNS = 32  # BASELINE configs[3]: the whole constellation of the strong-scaling leg

PRESETS = {
    1: {},
    2: dict(gnss="GPSL5", num_samples=50000, num_ants=4, num_taps=5, channels=12, blocks=1024),
    3: dict(gnss="GPSL1", num_samples=50000, num_ants=16, num_taps=3, channels=4, blocks=512),
    4: dict(gnss="GPSL1", num_samples=2000000, num_ants=64, num_taps=3, channels=64, blocks=1, block_ms=20.0),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle", type=int, default=64,
                    help="untimed launches before the warm-up steps (clock settle); reported in the JSON line")
    ap.add_argument("--blocks", type=int, default=4096, help="B: 1 ms integration blocks per launch")
    ap.add_argument("--num-samples", type=int, default=20000)
    ap.add_argument("--num-ants", type=int, default=4)
    ap.add_argument("--num-taps", type=int, default=3)
    ap.add_argument("--channels", type=int, default=1, help="K: PRN channels per GPU")
    ap.add_argument("--gnss", default="GPSL1")
    ap.add_argument("--layout", choices=["planar", "interleaved", "i16", "i8"], default="planar",
                    help="sample format: planar/interleaved ComplexF32 (headline), int16 / int8 ingest")
    ap.add_argument("--atomic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-block", action="store_true", help="skip the one-block-per-call latency figures")
    ap.add_argument("--no-shard-config3", action="store_true", help="N > 1: skip the configs[3] shard measurement")
    ap.add_argument("--constellation", action="store_true",
                    help="also measure BASELINE configs[3] as a whole (32 PRNs x 16 antennas @ 50 MHz, the PRNs sharded over the "
                         "ranks: strong scaling); on by default for N > 1")
    ap.add_argument("--no-constellation", action="store_true", help="N > 1: skip the strong-scaling constellation measurement")
    ap.add_argument("--no-read-ceiling", action="store_true", help="skip the in-run read-ceiling probe (roofline.read_ceiling_GBps)")
    ap.add_argument("--no-group-check", action="store_true",
                    help="N > 1: skip the device-group self-check (build/gat_multi_gpu in a fresh process after the ranks are done)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--block-ms", type=float, default=1.0, help="duration of one integration block (fs = N / block)")
    ap.add_argument("--ant-pad", type=int, default=0,
                    help="extra samples between the antennas' streams in the synthetic stream (experiments on memory-channel "
                         "mapping; the default 0 is the contiguous [M x B*N] layout)")
    ap.add_argument("--cpu-sweep", metavar="OUT.json", default=None,
                    help="CPU-baseline leg only (no GPU): time the oracle's FP32 4-pass port, 1 thread, on the reference's "
                         "single-block sweep grid (scripts/run_benchmarks_gpsl1.jl / _gpsl5.jl with processor = CPU) and "
                         "write the rows to OUT.json -- the CPU series scripts/plot_benchmarks.py draws beside the GPU's")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="launch-geometry option of the library (include/gat.h gat_set_option), repeatable; recorded in the line")
    ap.add_argument("--matrix-core", type=int, choices=[0, 1, 2, 3], default=None,
                    help="kernel selection (include/gat.h GAT_MC_*): 0 vector kernel only, 1 auto (default), 2 f32 MFMA, 3 split-bf16 MFMA")
    ap.add_argument("--baseline-config", type=int, choices=[1, 2, 3, 4], default=None,
                    help="shape of BASELINE.json configs[i] (1 = the default headline workload; 2 = GPS L5, 4 ants, 12 PRNs, "
                         "5 taps @ 50 MHz; 3 = the per-GPU shard of 16 ants x 32 PRNs @ 50 MHz; 4 = 64 ants x 64 channels, "
                         "20 ms @ 100 MHz); explicit shape flags still override nothing -- the preset wins")
    args = ap.parse_args(argv)
    if args.baseline_config is not None:
        for k, v in PRESETS[args.baseline_config].items():
            setattr(args, k, v)
        # (round 3: the presets used to cut settle / warm-up / steps to 4 / 3 / 20 "to keep the run short"; that measured the
        # first 18 ms after an idle device -- 0.667 ms per launch at the configs[3] shard where the steady state, from ~100
        # launches on, is 0.58 ms: profiles/r03/r03c_power_probe.txt.  All shapes now get the default settle + warm-up.)
    return args


# ----------------------------------------------------------------------------------------------
# self-launch: python bench.py --gpus N without an external launcher
# ----------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def group_check(members: int | None = None, timeout: float = 300.0) -> dict:
    """The device-group path of the C ABI over every visible device, in a FRESH child process (this process is never
    replaced, and the child starts from a program that has not touched the GPU): examples/gat_multi_gpu.c -- ingest on
    device 0, peer replication, sharded launch, gather, bit-comparison with shard-by-shard on device 0.  Its last output
    line is the JSON object returned here.  ``members``: member count when it differs from the device count (the
    one-GPU rehearsal puts two members on device 0).  GAT_BENCH_GROUP_CHECK_CMD replaces the command (tests without a GPU)."""
    import shlex

    cmd = os.environ.get("GAT_BENCH_GROUP_CHECK_CMD")
    argv = shlex.split(cmd) if cmd else [os.path.join(ROOT, "build", "gat_multi_gpu")] + ([str(members)] if members else [])
    try:
        p = subprocess.run(argv, capture_output=True, text=True, timeout=timeout)
    except (OSError, subprocess.TimeoutExpired) as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "cmd": " ".join(argv)}
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    try:
        rec = json.loads(lines[-1])
    except (IndexError, ValueError):
        return {"error": "no JSON line", "rc": p.returncode, "stdout": p.stdout[-400:], "stderr": p.stderr[-400:]}
    rec["rc"] = p.returncode
    rec["cmd"] = " ".join(os.path.relpath(a, ROOT) if os.path.isabs(a) else a for a in argv)
    return rec


def self_launch(n: int, want_group_check: bool) -> int:
    """Start n ranks of this script as CHILD processes (no GPU call has happened in this process, and it
    is never replaced by exec), take rank 0's line, add the device-group self-check (a further child process, run after
    every rank has exited and released its device) and print the line; return the worst exit status."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GAT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    # rank 0 prints one line at its very end: a reader thread keeps the pipe drained whatever else it writes
    import threading

    rank0_out: list[str] = []
    reader = threading.Thread(target=lambda: rank0_out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # poll ALL children: the first one that fails takes the others down with it (they would otherwise sit in the
    # rendezvous / barrier until the deadline), and so does the deadline.  Only the exact children started above.
    rc = 0
    deadline = time.time() + float(os.environ.get("GAT_BENCH_LAUNCH_TIMEOUT", "1500"))
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if live and (rc != 0 or time.time() > deadline):
            if rc == 0:
                rc = 124
            for p in live:
                p.terminate()
            t_kill = time.time() + 5.0
            for p in live:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            live = []
        elif live:
            time.sleep(0.05)
    reader.join(timeout=10.0)
    text = rank0_out[0] if rank0_out else ""
    for line in text.splitlines():
        if rc == 0 and line.strip().startswith("{"):
            try:
                rec = json.loads(line)
            except ValueError:
                print(line, flush=True)
                continue
            if want_group_check and rec.get("n_gpus", 1) > 1:
                share = os.environ.get("GAT_BENCH_SHARE_GPU") == "1"
                rec["group_check"] = group_check(rec["n_gpus"] if share else None)
            print(json.dumps(rec), flush=True)
        else:
            print(line, flush=True)
    return rc


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, host_re, host_im, prm, shifts, fs, system):
    """Time the oracle's FP32 4-pass CPU path on the first blocks of the SAME stream."""
    import oracle  # test infrastructure: used here only as the reported CPU baseline

    N, M = args.num_samples, args.num_ants
    codes = system.codes
    oprm = oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"],
                              prm["code_phase_chips"], prm["carrier_phase_cycles"])
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass

    max_blk = host_re.shape[1] // N

    def run(nblk, threads, budget):
        """Repeat passes over the first nblk blocks until `budget` seconds are used."""
        # the whole (contiguous) host arrays go in and the parameter slice says how many blocks are processed: a column
        # slice of the [M, B*N] arrays would be copied by the binding on every call (round 2 timed that copy with the
        # one-thread figure)
        oracle.dc_f32(host_re, host_im, codes, oprm[:min(4, nblk)], fs, shifts, N=N, threads=threads, native=True)  # warm-up
        done, t0 = 0, time.perf_counter()
        while True:
            oracle.dc_f32(host_re, host_im, codes, oprm[:nblk], fs, shifts, N=N, threads=threads, native=True)
            done += nblk
            dt = time.perf_counter() - t0
            if dt >= budget:
                return done, dt

    cands = sorted({t for t in (1, 4, 16, 64, cores) if t <= cores})
    rates = {}
    for t in cands:
        n, dt = run(min(64, max_blk) if t == 1 else max_blk, t, args.cpu_seconds / len(cands))
        rates[t] = n * N * args.channels / dt / 1e6
    used = max(rates, key=rates.get)
    best, rate_1t = rates[used], rates[1]
    # one thread, per pass (code replica / carrier replica / downconvert / correlate): where the 4-pass structure spends
    # its time on this host, in microseconds per (block, channel)
    nprof = min(64, max_blk)
    oracle.dc_f32_profile(host_re, host_im, codes, oprm[:nprof], fs, shifts, N=N, native=True)
    _, secs = oracle.dc_f32_profile(host_re, host_im, codes, oprm[:nprof], fs, shifts, N=N, native=True)
    per_pass = {k: round(float(v) / (nprof * args.channels) * 1e6, 3)
                for k, v in zip(("code_replica", "carrier_replica", "downconvert", "correlate"), secs)}
    return {
        "value": round(best, 3), "unit": "Msamples/s", "cores": used, "kind": "port",
        "sample": f"first {max_blk} of {args.blocks} blocks of the same stream, repeated for {args.cpu_seconds / len(cands):.1f} s per thread count "
                  f"(N={N}, M={M}, L={args.num_taps}, K={args.channels}); oracle FP32 4-pass (integer code / carrier NCOs, "
                  f"all four passes vectorised), gcc -O3 -march=native, OpenMP over (block, channel)",
        "per_pass_us": per_pass,
        "value_1_thread": round(rate_1t, 3), "by_threads": {str(k): round(v, 3) for k, v in rates.items()},
        "host_threads": cores, "cpu_model": cpu_model(),
    }


def cpu_sweep(out_path: str, seconds: float = 0.25):
    """The reference's committed sweep drives "processor" => ["CPU"] (scripts/run_benchmarks_gpsl1.jl:6; CPU branch
    src/benchmarks.jl:35-80: Tracking.downconvert_and_correlate! on ONE thread): the same grid here on the oracle's
    FP32 4-pass port (the reported CPU baseline; the reference's Julia path cannot run), prn 1, 1500 Hz, phases 0,
    0.5-chip spacing, minimum over repeated calls as BenchmarkTools' "Minimum"."""
    import oracle  # CPU-baseline leg: the only use of the oracle in this file besides the parity spot check

    grids = [("GPSL1", [2 ** e for e in range(11, 19)], [1, 4], [3, 7]), ("GPSL5", [2 ** e for e in range(15, 19)], [1, 4], [3])]
    rows = []
    for gnss, ns, ms, ls in grids:
        lc, fc, _ = oracle.SYSTEMS[gnss]
        codes = oracle.codes(gnss, 1)
        for n in ns:
            fs = n / 1e-3
            for m in ms:
                re, im = oracle.gen_signal(codes, 0, fc, fs, 1500.0, 0.0, 0.0, n, m)
                for l in ls:
                    sh = oracle.sample_shifts(l, fs, fc)
                    # timed inside C, call by call (a ctypes round trip per sample would add ~10 us of Python to a 5 us block)
                    per = oracle.dc_f32_time(re, im, codes, 0, fc, fs, 1500.0, 0.0, 0.0, sh, 20)[0].min() * 1e-9
                    reps = int(min(10000, max(20, seconds / max(per, 1e-7))))
                    times = oracle.dc_f32_time(re, im, codes, 0, fc, fs, 1500.0, 0.0, 0.0, sh, reps)[0]
                    t = np.asarray(times, dtype=np.float64)
                    rows.append({"processor": "CPU", "GNSS": gnss, "num_samples": n, "num_ants": m, "num_correlators": l,
                                 "algorithm": "cpu_port_1_thread", "Minimum": float(t.min()), "Median": float(np.median(t)),
                                 "Mean": float(t.mean()), "samples": int(t.size), "CPU_model": cpu_model(),
                                 "real_time_factor": 1e-3 / (t.min() * 1e-9)})
                    print(f"CPU {gnss} N={n:7d} M={m} L={l}  min {t.min() / 1e3:9.2f} us  RTF {1e-3 / (t.min() * 1e-9):8.2f}", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rows, f, indent=1)
    return rows


def algorithmic_flops(B, N, M, L, K) -> float:
    """per (n, k) carrier phase + sincos ~ 30; per (n, m, k) complex x complex = 8;
    per (n, m, k, l) +-1 complex MAC = 4."""
    return float(B) * N * K * (30.0 + M * (8.0 + 4.0 * L))


def step_stats(laps_ms) -> dict:
    """min / median / mean / sigma / max of the timed steps (BenchmarkTools' estimators, src/benchmarks.jl:1-9), in ms."""
    t = np.asarray(laps_ms, dtype=np.float64)
    if t.size == 0:
        return {"n": 0}
    return {"min": round(float(t.min()), 6), "median": round(float(np.median(t)), 6), "mean": round(float(t.mean()), 6),
            "sigma": round(float(t.std(ddof=1)) if t.size > 1 else 0.0, 6), "max": round(float(t.max()), 6), "n": int(t.size)}


def roofline(shape, launch_s, matrix_core, stored, mean_s=None, ceiling=None, bf16_terms=3):
    """max(bytes / HBM peak, flops / compute peak) vs the measured launch duration (the MEDIAN step; `mean_s`: the mean of
    the same steps beside it); names the winning term.  `stored`: (HBM bytes per launch, vector instructions per launch,
    where they come from) of the committed PMC passes; `ceiling`: the in-run read-ceiling record."""
    import gpuacceleratedtracking_amd as g

    B, N, M, L, K, layout = shape
    alg_bytes = g.algorithmic_bytes(B, N, M, L, K, g.SAMPLE_BYTES[layout])
    flops = algorithmic_flops(B, N, M, L, K)
    t_hbm = alg_bytes / (HBM_PEAK_GBS * 1e9)
    t_f32 = flops / (F32_PEAK_TFLOPS * 1e12)
    traffic, valu, traffic_source = stored
    terms = {"hbm": round(t_hbm * 1e3, 6), "f32_flops": round(t_f32 * 1e3, 6)}
    if valu and not matrix_core:
        # issue-rate term: a SIMD issues one wave64 vector instruction per 2 cycles (MI355X_MICROARCH.md: SIMD-32); the
        # launch's SQ_INSTS_VALU (stored PMC pass of this workload) over 1024 SIMDs at the 2.4 GHz the flop roof assumes
        terms["valu_issue"] = round(valu * VALU_CYCLES / (NUM_SIMDS * PEAK_CLOCK_HZ) * 1e3, 6)
    out = {"algorithmic_bytes_per_launch": alg_bytes, "algorithmic_flops_per_launch": flops,
           "kernel_ms_per_launch": round(launch_s * 1e3, 6), "kernel_ms_statistic": "median of the timed steps",
           "traffic": traffic, "traffic_source": traffic_source,
           "terms_ms": terms, "limiting_term": max(terms, key=terms.get),
           "kernel": {0: "dc_kernel (vector)", 1: "mfma_kernel (f32 MFMA)", 2: "mfma_bf16_kernel (split-bf16 MFMA)"}.get(matrix_core, "?")}
    if valu and not matrix_core:
        out["valu_insts_per_launch"] = valu
        out["frac_of_limiting_term"] = round(max(terms.values()) * 1e-3 / launch_s, 4)
    if t_hbm >= t_f32:
        ach = alg_bytes / launch_s / 1e9
        out.update(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))
    else:
        # compute roof: the f32 rate (vector FP32 == f32 MFMA, 157.3 TFLOP/s) bounds every formulation with f32
        # accuracy at the algorithmic flop count; the split-bf16 kernel issues 8x the flops on the bf16 pipe
        ach = flops / launch_s / 1e12
        out.update(bound="mfma" if matrix_core else "vector-f32", achieved=round(ach, 2), peak=F32_PEAK_TFLOPS,
                   unit="TFLOP/s", frac=round(ach / F32_PEAK_TFLOPS, 4))
        if matrix_core == 2:
            # k-slots the kernel issues per f32 product: float samples 3 terms x 3 terms less the lo x lo product = 8; int16
            # samples 2 exact terms: {a h, a m, a l, b h, b m} = 5; int8 samples 1 term: 3 + one zero slot = 4
            slots = {1: 4, 2: 5}.get(bf16_terms, 8)
            out["bf16_issue"] = {"executed_tflops": round(slots * 2.0 * 2 * M * 2 * K * L * N * B / launch_s / 1e12, 1),
                                 "peak": BF16_PEAK_TFLOPS, "slots_per_product": slots, "terms_per_sample": bf16_terms or 3,
                                 "frac": round(slots * 2.0 * 2 * M * 2 * K * L * N * B / launch_s / 1e12 / BF16_PEAK_TFLOPS, 4),
                                 "note": "split-bf16: `slots` bf16 products per f32 product, 2M x 2KL x N real GEMM"}
    out["hbm_frac"] = round(alg_bytes / launch_s / 1e9 / HBM_PEAK_GBS, 4)
    if mean_s:
        out["frac_mean"] = round(out["frac"] * launch_s / mean_s, 4)
        out["kernel_ms_per_launch_mean"] = round(mean_s * 1e3, 6)
    if ceiling:
        out["read_ceiling_GBps"] = ceiling.get("GBps")
        out["read_ceiling"] = ceiling
        if ceiling.get("GBps"):
            out["frac_of_ceiling"] = round(alg_bytes / launch_s / 1e9 / ceiling["GBps"], 4)
    return out


def read_ceiling(ctx, sig, layout_planar: bool) -> dict:
    """"also report vs the measured read ceiling".  A kernel that ONLY reads (gat_debug_read_stream:
    one 16-byte load per lane and step, nothing else) over the stream the correlator has just walked -- for the planar
    layout its re plane (half the stream, still far beyond the 256 MB Infinity Cache) --, every reader variant 6 launches,
    the best variant's median: what this device's memory system gives a read-once kernel in this process, now."""
    try:
        t = sig[0]
        nbytes = (t.numel() * t.element_size()) // 16 * 16
        best = None
        for variant in (0, 1, 2, 3, 4, 5, 8, 9):  # 8 .. 64 workgroups per CU x {nt, plain} x {8, 4} loads in flight
            ms = ctx.read_stream_ms(t, nbytes, variant=variant, launches=6)
            med = float(np.median(ms[1:]))
            if best is None or med < best[1]:
                best = (variant, med, float(ms[1:].min()))
        v, med, mn = best
        ms = ctx.read_stream_ms(t, nbytes, variant=v, launches=12)  # the winner once more, longer
        med, mn = float(np.median(ms[1:])), float(ms[1:].min())
        return {"GBps": round(nbytes / med / 1e6, 1), "GBps_best_launch": round(nbytes / mn / 1e6, 1), "bytes": int(nbytes),
                "ms_median": round(med, 6), "variant": v, "launches": 11,
                "what": "gat_debug_read_stream over the " + ("re plane of the " if layout_planar else "") + "same stream: 16-byte loads only"}
    except Exception as exc:  # noqa: BLE001 -- never fails the line
        return {"error": f"{type(exc).__name__}: {exc}"}


def stored_traffic(key):
    """HBM bytes and vector instructions per launch from the committed PMC passes (profiles/pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE, --pmc WRITE_SIZE and the SQ set in separate runs, FETCH_SIZE doubled per the microarchitecture guide),
    keyed by workload.  NOT counters of this run (a PMC pass needs the profiler around the process): returns (bytes,
    SQ_INSTS_VALU, where they come from)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(tpath) as f:
            tj = json.load(f)
        for e in tj.get("entries", [tj]):
            if e.get("workload_key") == key:
                return (e.get("hbm_bytes_per_launch"), e.get("valu_insts_per_launch"),
                        f"stored: profiles/pmc_traffic.json <- {tj.get('source', '?')}, kernel {e.get('kernel', '?')} "
                        f"(2 x FETCH_SIZE + WRITE_SIZE; SQ_INSTS_VALU; not measured in this run)")
    except Exception:
        pass
    return None, None, "none: no stored PMC pass for this workload"


def measure(args, g, torch, dist, world, rank, shape_kw, steps, warmup, settle, want_host_copy):
    """Build one stream + operator, run settle + warm-up + exactly `steps` timed launches between barriers."""
    layout = {"planar": g.GAT_LAYOUT_PLANAR, "interleaved": g.GAT_LAYOUT_INTERLEAVED,
              "i16": g.GAT_LAYOUT_INTERLEAVED_I16, "i8": g.GAT_LAYOUT_INTERLEAVED_I8}[shape_kw["layout"]]
    flags = g.GAT_FLAG_ATOMIC if args.atomic else 0
    N, M, L, K, B = (shape_kw[k] for k in ("num_samples", "num_ants", "num_taps", "channels", "blocks"))
    # channel sharding: rank r correlates PRNs [r*K, (r+1)*K) of the constellation on a replicated signal (weak scaling:
    # K per GPU whatever N is); with "channels_total" the constellation is fixed and ShardPlan cuts it (strong scaling)
    if shape_kw.get("channels_total"):
        plan = g.shard_channels(int(shape_kw["channels_total"]), world, rank)
        K = plan.count
        if K < 1:
            raise SystemExit(f"bench.py: rank {rank} of {world} gets no channel of {shape_kw['channels_total']}")
    else:
        plan = g.shard_channels(K * world, world, rank)
    op, desc, sig, prm = g.build_stream(shape_kw["gnss"], N, M, L, K, B, layout=layout, first_prn=plan.lo, flags=flags,
                                        block_seconds=shape_kw["block_ms"] * 1e-3, ant_pad=args.ant_pad)
    ctx = op.ctx
    for opt in args.option:
        name, _, val = opt.partition("=")
        ctx.set_option(name.strip(), int(val))
    if args.matrix_core is not None:
        ctx.set_matrix_core(args.matrix_core)
    fs = N / (shape_kw["block_ms"] * 1e-3)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock / TLB settle: the first few dozen launches after the stream has been synthesised run 3-6 % slower
    # (0.41-0.42 ms instead of 0.395 ms at configs[1]); like the data generation this is untimed set-up, ahead of
    # the contract's W warm-up steps, and its count is part of the printed record ("settle")
    for _ in range(settle):
        op.launch(desc)
    for _ in range(warmup):
        op.launch(desc)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    ctx.timer_lap()  # one HIP event per step on the launch stream: every step's duration, read after the timed region
    for _ in range(steps):
        op.launch(desc)
        ctx.timer_lap()
    kernel_ms_total = ctx.timer_stop()  # HIP events on the launch stream; synchronises
    barrier()
    elapsed = time.perf_counter() - t0
    laps = ctx.timer_laps(steps + 1)
    median_ms = float(np.median(laps)) if laps.size else kernel_ms_total / steps
    per_rank = [elapsed * 1e3 / steps]
    per_rank_kernel = [median_ms]
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    ident = {"rank": rank, "device": torch.cuda.current_device(), "name": ctx.device_info()["name"],  # name + gfx arch
             "pci_bus_id": "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                 getattr(props, "pci_device_id", 0)),
             "uuid": str(getattr(props, "uuid", ""))}
    idents = [ident]
    if dist is not None:
        dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        t = torch.tensor([elapsed, kernel_ms_total, median_ms], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total, median_ms = float(t[0]), float(t[1]), float(t[2])
        per_rank = [float(x[0]) * 1e3 / steps for x in gathered]
        per_rank_kernel = [float(x[2]) for x in gathered]
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    res = dict(devices=idents, op=op, desc=desc, sig=sig, prm=prm, ctx=ctx, fs=fs, layout=layout, shape=(B, N, M, L, K, layout),
               elapsed=elapsed, launch_s=kernel_ms_total * 1e-3 / steps, per_rank_ms=per_rank, steps=steps, laps_ms=laps,
               median_s=median_ms * 1e-3, per_rank_kernel_ms=per_rank_kernel, plan=plan, world=world)
    return res


def leg_record(g, m, gnss, layout_name, args, backend, want_ceiling):
    """What every measured shape reports beside its headline figures (rank 0): launch geometry, per-step statistics, roofline
    (median-based, with the read ceiling and the issue-rate term), parity spot check."""
    B, N, M, L, K, _ = m["shape"]
    info = m["ctx"].last_launch_info()
    err, h_re, h_im = parity_check(g, m)
    ceiling = read_ceiling(m["ctx"], m["sig"], layout_name == "planar") if want_ceiling else None
    rec = {"launch": info, "step_ms": step_stats(m["laps_ms"]),
           "roofline": roofline(m["shape"], m["median_s"], info.get("matrix_core", 0), stored_traffic([gnss, N, M, L, K, B, layout_name]), bf16_terms=info.get("bf16_terms", 3),
                                mean_s=m["launch_s"], ceiling=ceiling),
           "parity_max_rel_err_vs_f64_oracle": err}
    if m["world"] > 1:
        rec["ms_per_step_by_rank"] = [round(x, 6) for x in m["per_rank_ms"]]
        rec["kernel_ms_median_by_rank"] = [round(x, 6) for x in m["per_rank_kernel_ms"]]
        rec["backend"] = backend
    return rec, (h_re, h_im)


def parity_check(g, m):
    """Spot-check of the timed output against the FP64 oracle (first blocks; outside the timed region)."""
    import oracle

    B, N, M, L, K, layout = m["shape"]
    sig, op, prm = m["sig"], m["op"], m["prm"]
    host_blocks = min(B, 512)

    def host_copy(b0, nb):
        if layout == g.GAT_LAYOUT_PLANAR:
            return sig[0][:, b0 * N:(b0 + nb) * N].cpu().numpy(), sig[1][:, b0 * N:(b0 + nb) * N].cpu().numpy()
        h = sig[0][:, b0 * N:(b0 + nb) * N, :].cpu().numpy().astype(np.float32)  # ints convert exactly
        return np.ascontiguousarray(h[..., 0]), np.ascontiguousarray(h[..., 1])

    h_re, h_im = host_copy(0, host_blocks)
    got = op.result()
    oprm = oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"],
                              prm["code_phase_chips"], prm["carrier_phase_cycles"])
    nchk = min(2, host_blocks)
    err = 0.0
    for b0 in sorted({0, max(0, B - nchk)}):  # the first and the last blocks of the launch
        c_re, c_im = (h_re[:, :nchk * N], h_im[:, :nchk * N]) if b0 == 0 else host_copy(b0, nchk)
        ref = oracle.correlate_f64(c_re, c_im, op.system.codes, oprm[b0:b0 + nchk], m["fs"], op.shifts, N=N)
        err = max(err, float(np.max(np.abs(got[b0:b0 + nchk] - ref) / np.abs(ref).max(axis=(2, 3), keepdims=True))))
    return err, h_re, h_im


def single_block(g, args) -> dict:
    """What the reference's own benchmark times (src/benchmarks.jl:120-146): ONE 1 ms block of the headline shape per call,
    synchronised -- through an ordinary launch + wait (library-owned stream, completion flag) and rung into the resident
    correlator (gat_resident_*: no launch; outputs on the host when the call returns).  Python host layer, a fraction of a
    second; never fails the line (an error record instead).  examples/gat_latency.c is the native form over the whole grid."""
    try:
        from gpuacceleratedtracking_amd.algorithms import ALGODICT, KernelAlgorithm
        from gpuacceleratedtracking_amd.benchmarks import _run_kernel_benchmark

        N, M, L = args.num_samples, args.num_ants, args.num_taps
        rec = {"workload": f"{args.gnss}, {M} ants, 1 PRN, {L} correlators, one 1 ms block of {N} samples per call", "unit": "us",
               "host_layer": "python/ctypes"}
        res = {}
        for name in ("hip_fused", "hip_resident"):
            t, op, _ = _run_kernel_benchmark(g.GNSSDICT[args.gnss], N, M, L, KernelAlgorithm(ALGODICT[name]), seconds=0.15, max_samples=3000)
            res[name] = op.result()[0, 0]
            rec["launch_and_wait" if name == "hip_fused" else "resident_call"] = {
                "min": round(float(t.min()) / 1e3, 2), "median": round(float(np.median(t)) / 1e3, 2), "calls": int(t.size)}
            if name == "hip_resident":
                rec["resident_call"]["workgroups"] = op.resident_info["workgroups"]
        a, b = res["hip_fused"], res["hip_resident"]
        rec["resident_vs_launch_max_rel_diff"] = float(np.abs(a - b).max() / np.abs(a).max())
        try:  # the receiver loop with the host in it (gat_resident_tracking_run: resident call + loop filters on the CPU per block)
            rec["host_closed_loop"] = host_closed_loop(g, args)
        except Exception as exc:  # noqa: BLE001
            rec["host_closed_loop"] = {"error": f"{type(exc).__name__}: {exc}"}
        return rec
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def host_closed_loop(g, args, blocks: int = 400) -> dict:
    """`blocks` consecutive 1 ms blocks of the headline shape (one satellite on every antenna) through the native host-closed
    tracking loop: every block's correlator outputs and parameters are on the host.  Microseconds per block and real-time factor."""
    import time

    import torch
    system = g.GNSSDICT[args.gnss]()
    N, M, L = args.num_samples, args.num_ants, args.num_taps
    fs = N / 1e-3
    fc, dop, tau0 = g.get_code_frequency(system), 1200.0, 100.25
    fcode = fc * (1 + dop / 1575.42e6)
    b = np.arange(blocks, dtype=np.float64)[:, None]
    lc = g.get_code_length(system)
    prm = g.make_params(np.array([0]), np.array([fcode]), np.array([dop]), np.mod(tau0 + fcode * 1e-3 * b, lc), 2 * np.pi * np.mod(dop * 1e-3 * b, 1.0),
                        shape=(blocks, 1))
    re, im = g.gen_signal_stream(system, prm, fs, N, M)
    shifts = g.get_correlator_sample_shifts(system, g.EarlyPromptLateCorrelator(M, L), fs, 0.5)
    torch.cuda.synchronize()
    with g.ResidentTrackingLoop(system, np.array([1]), N, M, fs, shifts, np.array([dop + 5.0]), np.array([tau0 + 0.05]), re=re, im=im, idle_us=200000) as loop:
        loop.run(50)
        t0 = time.perf_counter()
        loop.run(blocks - 50, start=50 * N)
        dt = (time.perf_counter() - t0) / (blocks - 50)
        st, acc = loop.state(), loop.accumulators()
    prompt = float(np.abs(acc[0, L // 2, :]).min()) / N
    return {"us_per_block": round(dt * 1e6, 2), "real_time_factor": round(1e-3 / dt, 1), "blocks": blocks - 50, "workgroups": loop.resident_workgroups,
            "doppler_error_hz": round(float(abs(st["carrier_doppler_hz"][0] - dop)), 3), "prompt_over_N": round(prompt, 3)}


def main():
    args = parse_args()
    if args.cpu_sweep:
        cpu_sweep(args.cpu_sweep)
        return
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # no external launcher: become the launcher.  Nothing above has touched the GPU (torch is not even imported).
        sys.exit(self_launch(args.gpus, not args.no_group_check))

    import torch

    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("GAT_BENCH_DRYRUN") == "1":
        # launcher rehearsal without a GPU (tests/test_bench_launcher.py): rendezvous, barrier, gather, one line
        if os.environ.get("GAT_BENCH_DRYRUN_FAIL_RANK") == str(rank):  # rehearsal of a rank that dies before the rendezvous
            print(f"bench.py: rank {rank} fails on request", file=sys.stderr)
            sys.exit(3)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        t = torch.tensor([float(rank)], dtype=torch.float64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        # the strong-scaling leg's plan, computed by every rank for itself as measure() does, then gathered
        from gpuacceleratedtracking_amd.sharding import ShardPlan  # host arithmetic only (no GPU, no library load)
        mine = ShardPlan(CONSTELLATION_PRNS, world, rank)
        plans = [None] * world
        dist.all_gather_object(plans, (rank, mine.lo, mine.count))
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            rec = {"dryrun": True, "n_gpus": world, "ranks_seen": [int(x[0]) for x in got], "local_rank": local_rank,
                   "constellation_config3": {"prns_total": CONSTELLATION_PRNS, "first_prn_by_rank": [p[1] for p in sorted(plans)],
                                             "prns_by_rank": [p[2] for p in sorted(plans)], "scaling": "strong"}}
            if world > 1 and os.environ.get("GAT_BENCH_CHILD") != "1" and not args.no_group_check:
                rec["group_check"] = group_check()  # external launcher: rank 0 runs it once the group is gone
            print(json.dumps(rec), flush=True)
        return
    if not torch.cuda.is_available():
        print("bench.py: no HIP device", file=sys.stderr)
        sys.exit(2)
    # GAT_BENCH_SHARE_GPU=1: rehearsal of the N-rank path on a box with ONE GPU (every rank on device 0, gloo for the
    # barrier because RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("GAT_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    if dev_index >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} needs device {dev_index}, only {torch.cuda.device_count()} visible", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(dev_index)
    dist = None
    backend = None
    if world > 1 or os.environ.get("GAT_BENCH_FORCE_DIST") == "1":  # the latter: exercise the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if share else "nccl"
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend="gloo")

    import gpuacceleratedtracking_amd as g

    shape_kw = dict(gnss=args.gnss, num_samples=args.num_samples, num_ants=args.num_ants, num_taps=args.num_taps,
                    channels=args.channels, blocks=args.blocks, layout=args.layout, block_ms=args.block_ms)
    m = measure(args, g, torch, dist, world, rank, shape_kw, args.steps, args.warmup, args.settle, True)
    B, N, M, L, K, layout = m["shape"]
    fs = m["fs"]

    out = None
    want_ceiling = not args.no_read_ceiling
    if rank == 0:
        total_samples = float(B) * N * K * world * args.steps
        value = total_samples / m["elapsed"] / 1e6
        rec, (h_re, h_im) = leg_record(g, m, args.gnss, args.layout, args, backend, want_ceiling)
        out = {
            "metric": "Msamples/s downconvert+correlate (E/P/L x ants x sats); real-time factor @ 1ms",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "settle": args.settle, "ms_per_step": round(m["elapsed"] * 1e3 / args.steps, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.layout in ("planar", "interleaved") else f"f32 (samples {args.layout})",
            "data": "synthetic",
            "config": {
                "workload": f"{args.gnss}, {M} ants, {K} PRN/GPU, {L} correlators, {args.block_ms:g} ms @ {fs / 1e6:g} MHz "
                            f"{'(BASELINE configs[1]) ' if (args.gnss, N, M, L, K) == ('GPSL1', 20000, 4, 3, 1) else ''}"
                            f"batched stream of {B} blocks/launch",
                "num_samples": N, "num_ants": M, "num_taps": L, "channels_per_gpu": K, "blocks_per_launch": B,
                "layout": args.layout, "second_stage": "atomic" if args.atomic else "deterministic",
                "sharding": f"channels x{world} (replicated signal, no collective)",
                "launch": rec["launch"],
            },
            "real_time_factor": round(value * 1e6 / fs / (K * world), 3),
            "step_ms": rec["step_ms"],
            "roofline": rec["roofline"],
            "parity_max_rel_err_vs_f64_oracle": rec["parity_max_rel_err_vs_f64_oracle"],
        }
        from gpuacceleratedtracking_amd.benchmarks import provenance
        out.update(provenance())  # "libgat": version + kernel-source commit + build flags, "git": repository commit
        if args.option:
            out["config"]["options"] = list(args.option)
        if args.matrix_core is not None:
            out["config"]["options"] = out["config"].get("options", []) + [f"matrix_core={args.matrix_core}"]
        if world > 1:
            out["ranks"] = {"world_size": world, "backend": backend, "ms_per_step_by_rank": rec["ms_per_step_by_rank"],
                            "kernel_ms_median_by_rank": rec["kernel_ms_median_by_rank"], "devices": m["devices"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, h_re, h_im, m["prm"], m["op"].shifts, fs, m["op"].system)
        del h_re, h_im
    del m
    if rank == 0 and world == 1 and not args.no_single_block:
        torch.cuda.empty_cache()
        out["single_block"] = single_block(g, args)

    # N > 1: the shape the 8-GPU node of BASELINE configs[3] runs -- 16 antennas, 32 PRNs sharded 4 per GPU, 50 MHz
    if world > 1 and not args.no_shard_config3 and args.baseline_config is None:
        torch.cuda.empty_cache()
        kw3 = dict(PRESETS[3], layout="planar", block_ms=1.0)
        # the same protocol as the headline: this shape reads 0.70 ms per launch in the first dozens of launches after an idle
        # device and 0.58 ms in the steady state (profiles/r03/r03d_steady_state_by_warmup.txt) -- a shortened settle /
        # warm-up would publish the cold number
        steps3 = args.steps
        m3 = measure(args, g, torch, dist, world, rank, kw3, steps3, args.warmup, args.settle, False)
        if rank == 0:
            B3, N3, M3, L3, K3, _ = m3["shape"]
            v3 = float(B3) * N3 * K3 * world * steps3 / m3["elapsed"] / 1e6
            rec3, _ = leg_record(g, m3, "GPSL1", "planar", args, backend, False)
            out["shard_config3"] = {
                "workload": f"GPSL1, {M3} ants, {K3 * world} PRNs sharded {K3}/GPU over {world} GPUs, {L3} correlators, 1 ms @ "
                            f"{m3['fs'] / 1e6:g} MHz, {B3} blocks/launch (BASELINE configs[3]: 32 PRNs at 8 GPUs)",
                "value": round(v3, 3), "unit": "Msamples/s", "n_gpus": world, "scaling": "weak", "steps": steps3, "warmup": args.warmup,
                "settle": args.settle, "ms_per_step": round(m3["elapsed"] * 1e3 / steps3, 6),
                "real_time_factor": round(v3 * 1e6 / m3["fs"] / (K3 * world), 3),
                "rccl_world_size": world, **rec3,
            }
        del m3

    # BASELINE configs[3] AS A WHOLE on however many GPUs there are (strong scaling): 32 PRNs x 16 antennas @ 50 MHz, the PRNs
    # cut by ShardPlan(32, world, rank) -- 32 / 16 / 8 / 4 per GPU at N = 1 / 2 / 4 / 8 -- on a replicated signal, no collective
    # (the reference's several-satellites-per-launch form: src/algorithms.jl:637-718).  value = samples x 32 channels per second
    # of the whole receiver; real_time_factor = signal time per wall time of the slowest rank.
    if (args.constellation or world > 1) and not args.no_constellation and args.baseline_config is None:
        torch.cuda.empty_cache()
        kwc = dict(PRESETS[3], layout="planar", block_ms=1.0, channels_total=CONSTELLATION_PRNS)
        mc = measure(args, g, torch, dist, world, rank, kwc, args.steps, args.warmup, args.settle, False)
        if rank == 0:
            Bc, Nc, Mc, Lc_, Kc, _ = mc["shape"]
            vc = float(Bc) * Nc * CONSTELLATION_PRNS * args.steps / mc["elapsed"] / 1e6
            recc, _ = leg_record(g, mc, "GPSL1", "planar", args, backend, False)
            out["constellation_config3"] = {
                "workload": f"GPSL1, {Mc} ants, {CONSTELLATION_PRNS} PRNs over {world} GPU(s) = {mc['plan'].counts()} per GPU "
                            f"(ShardPlan, contiguous), {Lc_} correlators, 1 ms @ {mc['fs'] / 1e6:g} MHz, {Bc} blocks/launch "
                            f"(BASELINE configs[3] as a whole)",
                "value": round(vc, 3), "unit": "Msamples/s", "n_gpus": world, "scaling": "strong", "steps": args.steps,
                "warmup": args.warmup, "settle": args.settle, "ms_per_step": round(mc["elapsed"] * 1e3 / args.steps, 6),
                "real_time_factor": round(Bc * 1e-3 * args.steps / mc["elapsed"], 3),
                "prns_by_rank": mc["plan"].counts(), "rccl_world_size": world, **recc,
            }
        del mc
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and os.environ.get("GAT_BENCH_CHILD") != "1" and not args.no_group_check:
            # external launcher (the driver's torch.distributed.run form): the ranks are past their last barrier and
            # this one holds no buffer any more; the check runs in a fresh child process over every visible device
            # (under self-launch the parent, which never touched the GPU, does it after every rank has exited)
            torch.cuda.empty_cache()
            out["group_check"] = group_check(world if share else None)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
