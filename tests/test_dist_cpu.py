"""Multi-process logic of the batch-sharded sampler on CPU: gloo, world size 2 (the GPU path uses the same code with RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mm-diffusion_amd"))
    from mm_diffusion import dist_util
    dist_util.setup_dist(backend="gloo")
    assert dist_util.world_size() == world and dist_util.rank() == rank
    # shard a global batch of 5 -> [0,3) and [3,5); every sample id appears exactly once after the terminal all-gather
    lo, hi = dist_util.shard_batch(5)
    mine = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1, 1).expand(-1, 2, 3).contiguous()
    pad = torch.zeros(3 - (hi - lo), 2, 3)                  # all_gather needs equal shapes: pad to the largest shard
    gathered = dist_util.all_gather_samples(torch.cat([mine, pad]))
    ids = torch.cat([gathered[0:3, 0, 0], gathered[3:3 + 2, 0, 0]])
    ok_gather = ids.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
    # flat parameter broadcast: rank 1 starts with garbage and must end with rank 0's values
    torch.manual_seed(rank)
    params = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(7))]
    ref = [torch.randn(4, 3, generator=torch.Generator().manual_seed(0))]  # noqa: F841 (only to document intent)
    dist_util.sync_params(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    allf = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(allf, flat)
    ok_sync = all(torch.equal(allf[0], a) for a in allf)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok_max = float(t) == float(world)
    dist.barrier()
    q.put((rank, ok_gather, ok_sync, ok_max))
    dist.destroy_process_group()


def test_gloo_world2_shard_gather_sync():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True, True), (1, True, True, True)]
