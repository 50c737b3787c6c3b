#!/usr/bin/env python3
"""bench.py -- headline measurement of the downconvert + correlate hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the one the metric is quoted on): GPS L1 C/A, 4 antennas,
1 PRN per GPU, 3 E/P/L correlators, 1 ms integration blocks at fs = 20 MHz (N = 20 000), as a
batched stream of B consecutive blocks resident in HBM (B = 4096 -> 2.6 GB, far beyond the 256 MB
Infinity Cache).  One "step" = one pass of the fused kernel over all B blocks.

Prints ONE JSON line on rank 0:
  value      = total samples correlated per second over all ranks [Msamples/s]
               (inputs resident in HBM when the timed region starts; sync-inclusive)
  roofline   = algorithmic bytes per launch / mean launch duration (HIP events on the launch
               stream) against the 8 TB/s HBM3E peak
  cpu_baseline = the oracle's FP32 4-pass CPU restatement ("port") timed on this host on a bounded
               sample of the same stream (the reference's Julia CPU path cannot run here)
Multi-GPU: satellite channels shard with no collective; every rank holds the full antenna signal
and correlates its own PRN ("weak" scaling: per-GPU work fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle", type=int, default=64, help="untimed launches before the warm-up steps (clock settle)")
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--block-ms", type=float, default=1.0, help="duration of one integration block (fs = N / block)")
    ap.add_argument("--baseline-config", type=int, choices=[1, 2, 3, 4], default=None,
                    help="shape of BASELINE.json configs[i] (1 = the default headline workload; 2 = GPS L5, 4 ants, 12 PRNs, "
                         "5 taps @ 50 MHz; 3 = the per-GPU shard of 16 ants x 32 PRNs @ 50 MHz; 4 = 64 ants x 64 channels, "
                         "20 ms @ 100 MHz); explicit shape flags still override nothing -- the preset wins")
    args = ap.parse_args()
    presets = {
        1: {},
        2: dict(gnss="GPSL5", num_samples=50000, num_ants=4, num_taps=5, channels=12, blocks=1024),
        3: dict(gnss="GPSL1", num_samples=50000, num_ants=16, num_taps=3, channels=4, blocks=512),
        4: dict(gnss="GPSL1", num_samples=2000000, num_ants=64, num_taps=3, channels=64, blocks=1, block_ms=20.0),
    }
    if args.baseline_config is not None:
        for k, v in presets[args.baseline_config].items():
            setattr(args, k, v)
        if args.baseline_config != 1:  # long launches: fewer timed steps keep the run short
            args.steps, args.warmup, args.settle = min(args.steps, 20), min(args.warmup, 3), min(args.settle, 4)
    return args


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
        oracle.dc_f32(host_re[:, :min(4, nblk) * N], host_im[:, :min(4, nblk) * N], codes, oprm[:min(4, nblk)],
                      fs, shifts, N=N, threads=threads, native=True)  # warm-up: page faults, OpenMP pool
        done, t0 = 0, time.perf_counter()
        while True:
            oracle.dc_f32(host_re[:, :nblk * N], host_im[:, :nblk * N], codes, oprm[:nblk], fs, shifts, N=N,
                          threads=threads, native=True)
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
    return {
        "value": round(best, 3), "unit": "Msamples/s", "cores": used, "kind": "port",
        "sample": f"first {max_blk} of {args.blocks} blocks of the same stream, repeated for {args.cpu_seconds / len(cands):.1f} s per thread count "
                  f"(N={N}, M={M}, L={args.num_taps}, K={args.channels}); oracle FP32 4-pass, gcc -O3 -march=native, OpenMP",
        "value_1_thread": round(rate_1t, 3), "by_threads": {str(k): round(v, 3) for k, v in rates.items()},
        "host_threads": cores,
    }


def main():
    args = parse_args()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no HIP device", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("GAT_BENCH_FORCE_DIST") == "1":  # the latter: exercise the RCCL path on 1 GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import gpuacceleratedtracking_amd as g

    layout = {"planar": g.GAT_LAYOUT_PLANAR, "interleaved": g.GAT_LAYOUT_INTERLEAVED,
              "i16": g.GAT_LAYOUT_INTERLEAVED_I16, "i8": g.GAT_LAYOUT_INTERLEAVED_I8}[args.layout]
    flags = g.GAT_FLAG_ATOMIC if args.atomic else 0
    N, M, L, K, B = args.num_samples, args.num_ants, args.num_taps, args.channels, args.blocks
    # channel sharding: rank r correlates PRNs [r*K, (r+1)*K) of the constellation on a replicated signal
    plan = g.shard_channels(K * world, world, rank)
    op, desc, sig, prm = g.build_stream(args.gnss, N, M, L, K, B, layout=layout, first_prn=plan.lo, flags=flags,
                                        block_seconds=args.block_ms * 1e-3)
    ctx = op.ctx
    fs = N / (args.block_ms * 1e-3)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock / TLB settle: the first few dozen launches after the stream has been synthesised run 3-6 % slower
    # (0.41-0.42 ms instead of 0.395 ms at configs[1]); like the data generation this is untimed set-up, ahead of
    # the contract's W warm-up steps, so that a short K still measures the steady state
    for _ in range(args.settle):
        op.launch(desc)
    for _ in range(args.warmup):
        op.launch(desc)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        op.launch(desc)
    kernel_ms_total = ctx.timer_stop()  # HIP events on the launch stream; synchronises
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed, kernel_ms_total], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_total = float(t[0]), float(t[1])

    if rank == 0:
        total_samples = float(B) * N * K * world * args.steps
        value = total_samples / elapsed / 1e6
        launch_s = kernel_ms_total * 1e-3 / args.steps
        alg_bytes = g.algorithmic_bytes(B, N, M, L, K, g.SAMPLE_BYTES[layout])
        achieved = alg_bytes / launch_s / 1e9
        info = ctx.last_launch_info()
        # parity spot-check of the timed output against the FP64 oracle (first / last blocks)
        import oracle
        host_blocks = min(B, 512)
        if layout == g.GAT_LAYOUT_PLANAR:
            h_re = sig[0][:, :host_blocks * N].cpu().numpy()
            h_im = sig[1][:, :host_blocks * N].cpu().numpy()
        else:
            h = sig[0][:, :host_blocks * N, :].cpu().numpy().astype(np.float32)  # ints convert exactly
            h_re, h_im = np.ascontiguousarray(h[..., 0]), np.ascontiguousarray(h[..., 1])
        got = op.result()
        oprm = oracle.make_params(prm["prn"], prm["code_freq_hz"], prm["carrier_freq_hz"],
                                  prm["code_phase_chips"], prm["carrier_phase_cycles"])
        nchk = min(2, host_blocks)
        ref = oracle.correlate_f64(h_re[:, :nchk * N], h_im[:, :nchk * N], op.system.codes, oprm[:nchk], fs,
                                   op.shifts, N=N)
        err = float(np.max(np.abs(got[:nchk] - ref) / np.abs(ref).max(axis=(2, 3), keepdims=True)))
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("workload_key") == [args.gnss, N, M, L, K, B, args.layout]:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s downconvert+correlate (E/P/L x ants x sats); real-time factor @ 1ms",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 6),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.layout in ("planar", "interleaved") else f"f32 (samples {args.layout})",
            "data": "synthetic",
            "config": {
                "workload": f"{args.gnss}, {M} ants, {K} PRN/GPU, {L} correlators, {args.block_ms:g} ms @ {fs / 1e6:g} MHz "
                            f"{'(BASELINE configs[1]) ' if (args.gnss, N, M, L, K) == ('GPSL1', 20000, 4, 3, 1) else ''}"
                            f"batched stream of {B} blocks/launch",
                "num_samples": N, "num_ants": M, "num_taps": L, "channels_per_gpu": K, "blocks_per_launch": B,
                "layout": args.layout, "second_stage": "atomic" if args.atomic else "deterministic",
                "sharding": f"channels x{world} (replicated signal, no collective)",
                "launch": info,
            },
            "real_time_factor": round(value * 1e6 / fs / (K * world), 3),
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms_per_launch": round(launch_s * 1e3, 6),
            },
            "parity_max_rel_err_vs_f64_oracle": err,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, h_re, h_im, prm, op.shifts, fs, op.system)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
