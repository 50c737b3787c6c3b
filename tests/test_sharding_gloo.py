"""world_size-2 gloo test of the N > 1 path on CPU: channel sharding + control-plane gather +
max-over-ranks timing.  The per-rank compute is stood in by the oracle (no GPU here); on the GPU
box the same ShardPlan drives bench.py's ranks."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import oracle
from tests.helpers import make_case, oracle_result


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    import gpuacceleratedtracking_amd as g
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = make_case(4242, N=3000, M=2, L=3, K=5, B=2)  # every rank: same replicated signal
        plan = g.shard_channels(case["K"], world, rank)
        prm = case["prm"][:, plan.lo:plan.hi]
        local = oracle.correlate_f64(case["re"], case["im"], case["codes"], np.ascontiguousarray(prm), case["fs"],
                                     case["shifts"], N=case["N"])
        assert local.shape[1] == plan.count
        full = g.gather_outputs(local, plan)
        t = g.sharding.max_over_ranks(float(rank + 1))
        if rank == 0:
            ret["full"] = full
            ret["tmax"] = t
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_channel_sharding_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    case = make_case(4242, N=3000, M=2, L=3, K=5, B=2)
    ref = oracle_result(case)
    assert ret["full"].shape == ref.shape
    assert np.array_equal(ret["full"], ref)  # disjoint channels: concatenation is exact
    assert ret["tmax"] == 2.0
