#!/bin/bash
# Round 5: bench.py's launch forms on the 1-GPU box with the final kernels.
set -o pipefail
out=gpurun_out/r05z; mkdir -p $out
# (0) the driver's N = 1 form
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default_driver_form.json 2> $out/bench_default_driver_form.err; echo "default rc=$?"
# (1) plain `python bench.py --gpus 2`: bench.py starts the two ranks itself (both on device 0, gloo for the barrier), measures
#     configs[1] and the configs[3] shard, then runs the device-group self-check (two members on device 0) in a fresh process
GAT_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_2ranks_one_gpu_rehearsal.json 2> $out/bench_2ranks_one_gpu_rehearsal.err; echo "selflaunch rc=$?"
# (2) the external-launcher form with two ranks on the one device (the driver's torch.distributed.run form; rank 0 runs the check)
GAT_BENCH_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29778 bench.py --gpus 2 --steps 20 --warmup 5 > $out/bench_torchrun_2ranks_one_gpu.json 2> $out/bench_torchrun_2ranks_one_gpu.err; echo "torchrun2 rc=$?"
# (3) the RCCL path with one rank (process group on the nccl backend, barrier + all_reduce on the device)
GAT_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_rccl_path_1rank.json 2> $out/bench_rccl_path_1rank.err; echo "force_dist rc=$?"
# (4) external launcher form, world size 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_torchrun_form_1rank.json 2> $out/bench_torchrun_form_1rank.err; echo "torchrun rc=$?"
for f in $out/*.json; do python - $f <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
d=json.loads(lines[-1]); r=d["roofline"]
print("%-48s n_gpus %d value %.0f ms/step %.4f frac %.4f libgat %s" % (sys.argv[1].split("/")[-1], d["n_gpus"], d["value"], d["ms_per_step"], r["frac"], d.get("libgat")))
if "shard_config3" in d: s=d["shard_config3"]; print("    shard_config3: ms/step %.4f steps %d warmup %d settle %d frac %.4f" % (s["ms_per_step"], s["steps"], s["warmup"], s["settle"], s["roofline"]["frac"]))
if "constellation_config3" in d: s=d["constellation_config3"]; print("    constellation_config3: %s | ms/step %.4f RTF %.1f scaling %s prns_by_rank %s by_rank %s frac %.4f err %.2e" % (s["workload"][:60], s["ms_per_step"], s["real_time_factor"], s["scaling"], s["prns_by_rank"], s.get("ms_per_step_by_rank"), s["roofline"]["frac"], s["parity_max_rel_err_vs_f64_oracle"]))
if "step_ms" in d: print("    step_ms:", d["step_ms"], "ceiling", r.get("read_ceiling_GBps"), "frac_of_ceiling", r.get("frac_of_ceiling"))
if "group_check" in d: print("    group_check:", {k: d["group_check"].get(k) for k in ("devices","members","peers_on_other_devices","bit_identical","peer_copy_GBps","rc","error")})
if "ranks" in d: print("    ranks:", [(x["rank"], x["name"], x["pci_bus_id"]) for x in d["ranks"]["devices"]], d["ranks"]["backend"])
if "cpu_baseline" in d: c=d["cpu_baseline"]; print("    cpu_baseline: %.0f Msamples/s on %d threads, %.0f on 1 (%s)" % (c["value"], c["cores"], c["value_1_thread"], c["cpu_model"]))
PY
done
