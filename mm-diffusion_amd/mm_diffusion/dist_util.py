"""Process-group bootstrap for one process per GPU over RCCL/xGMI (torchrun env rendezvous).

The reference bootstraps with mpi4py + CUDA_VISIBLE_DEVICES (dist_util.py:18-49); here rank / world size /
master address come from the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and
the backend is "nccl" (= RCCL on ROCm) when a GPU is visible, "gloo" otherwise (CPU tests)."""
import os

import torch as th
import torch.distributed as dist


def setup_dist(devices=None, backend=None):
    if dist.is_initialized():
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if th.cuda.is_available():
        th.cuda.set_device(local % th.cuda.device_count())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    # MMD_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices): used to smoke-test the multi-rank
    # code path of bench.py on a single-GPU box
    backend = backend or os.environ.get("MMD_DIST_BACKEND") or ("nccl" if th.cuda.is_available() else "gloo")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)


def dev():
    if th.cuda.is_available():
        return th.device("cuda", th.cuda.current_device())
    return th.device("cpu")


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def load_state_dict(path, **kwargs):
    return th.load(path, **kwargs)


def sync_params(params):
    """Rank-0 parameters to every rank as ONE flat broadcast (the reference issues one dist.broadcast per
    tensor: 1046 calls, dist_util.py:72-78)."""
    params = [p for p in params]
    if not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return
    with th.no_grad():
        flat = th.cat([p.detach().reshape(-1).float() for p in params])
        dist.broadcast(flat, 0)
        off = 0
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p).to(p.dtype))
            off += n


def shard_batch(global_batch, world=None, rnk=None):
    """Batch-sharded sampling: contiguous per-rank slice [lo, hi) of the global batch (independent trajectories,
    no in-loop communication - reference sample_sr.py:101-258 runs one replica per GPU)."""
    world = world_size() if world is None else world
    rnk = rank() if rnk is None else rnk
    base, extra = divmod(global_batch, world)
    lo = rnk * base + min(rnk, extra)
    return lo, lo + base + (1 if rnk < extra else 0)


def all_gather_samples(t):
    """Terminal all-gather of per-rank samples (reference mtu:424-431), one collective per tensor."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    out = [th.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous())
    return th.cat(out, dim=0)
