#!/bin/bash
# Round 2: rehearse bench.py's launch paths on the 1-GPU box and take baseline numbers of the unchanged kernels.
set -o pipefail
mkdir -p gpurun_out/r02a
# (1) plain `python bench.py --gpus 2`: bench.py starts the two ranks itself; both on device 0, gloo for the barrier
GAT_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --blocks 1024 > gpurun_out/r02a/selflaunch_2ranks_shared_gpu.json 2> gpurun_out/r02a/selflaunch_2ranks_shared_gpu.err
echo "selflaunch rc=$?"
# (2) the RCCL path with one rank (process group on the nccl backend, barrier + all_reduce on the device)
GAT_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/force_dist_rccl_1rank.json 2> gpurun_out/r02a/force_dist_rccl_1rank.err
echo "force_dist rc=$?"
# (3) external launcher form, as the driver calls it, world size 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/torchrun_1rank.json 2> gpurun_out/r02a/torchrun_1rank.err
echo "torchrun rc=$?"
for c in 2 3 4; do
  timeout -k 10 300 python bench.py --baseline-config $c --no-cpu-baseline > gpurun_out/r02a/baseline_c$c.json 2> gpurun_out/r02a/baseline_c$c.err
  echo "config $c rc=$?"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --num-samples 4000 --num-ants 1 --blocks 16384 > gpurun_out/r02a/baseline_c1shape.json 2> gpurun_out/r02a/baseline_c1shape.err
echo "c1 rc=$?"
tail -c 600 gpurun_out/r02a/*.err
