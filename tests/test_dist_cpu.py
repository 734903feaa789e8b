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


def _train_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mm-diffusion_amd"))
    from mm_diffusion import dist_util
    from mm_diffusion.optim import FlatAdamW
    from mm_diffusion.resample import LossSecondMomentResampler
    dist_util.setup_dist(backend="gloo")
    # data-parallel gradient reduction of the training step: ONE all-reduce of the flat gradient buffer -> the mean
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(11))]
    opt = FlatAdamW(params, lr=1e-3)
    assert params[0].data_ptr() == opt.flat.data_ptr()                 # parameters re-homed into the flat buffer
    loss = sum(((p * (rank + 1)) ** 2).sum() for p in params)          # rank-dependent gradient: 2 (rank+1)^2 p
    loss.backward()
    assert params[1].grad.data_ptr() == opt.grad[15:].data_ptr()       # autograd accumulated in place
    opt.all_reduce_grads()
    expect = torch.cat([p.detach().reshape(-1) for p in params]) * 2 * (1 + 4) / 2
    ok_grad = torch.allclose(opt.grad, expect, rtol=1e-6)

    # loss-aware timestep sampler: every rank ends with the same history although each contributed different pairs
    class D:
        num_timesteps = 6
    s = LossSecondMomentResampler(D(), history_per_term=2)
    for it in range(2):
        ts = torch.tensor([0, 1, 2] if rank == 0 else [3, 4, 5, 5][:3 + it])
        s.update_with_local_losses(ts, (ts.float() + 1) * (it + 1))
    hist = torch.from_numpy(s._loss_history.copy())
    allh = [torch.empty_like(hist) for _ in range(world)]
    dist.all_gather(allh, hist)
    ok_hist = all(torch.equal(allh[0], h) for h in allh) and float(hist[5].sum()) > 0 and float(hist[0].sum()) > 0
    dist.barrier()
    q.put((rank, ok_grad, ok_hist))
    dist.destroy_process_group()


def test_gloo_world2_training_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mm-diffusion_amd"))
    from mm_diffusion import dist_util
    from mm_diffusion.optim import FlatAdamW
    dist_util.setup_dist(backend="gloo")
    shapes = [(7, 3), (5,), (16, 4), (9,), (3, 3, 3), (11,), (2, 8)]
    res = []
    for buckets, overlap in ((1, False), (3, False), (3, True), (7, True)):
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        opt = FlatAdamW(params, lr=1e-3, grad_buckets=buckets)
        assert 1 <= len(opt.buckets) <= min(buckets, len(params)) and opt.buckets[0][0] == 0 and opt.buckets[-1][1] == opt.grad.numel()
        assert sum(b[2] for b in opt.buckets) == len(params) and all(a[1] == b[0] for a, b in zip(opt.buckets, opt.buckets[1:]))
        assert buckets == 1 or len(opt.buckets) > 1
        g = torch.Generator().manual_seed(100 + rank)
        opt.grad.copy_(torch.randn(opt.grad.numel(), generator=g))
        expect = opt.grad.clone()
        dist.all_reduce(expect, op=dist.ReduceOp.SUM)              # the single flat all-reduce the bucketed one must equal bitwise
        expect.div_(world)
        if overlap:
            opt.arm_overlap()
            for i in reversed(range(len(params))):                 # backward order: last parameter first
                opt._flush_ready()                                 # what train_ops._grad_slot does per backward kernel
                opt._param_done(i)
            opt._flush_ready()
            launched_early = len(opt._inflight)
        else:
            launched_early = 0
        opt.all_reduce_grads()
        res.append((torch.equal(opt.grad, expect), launched_early == (len(opt.buckets) if overlap else 0)))
        assert not opt._inflight and not opt._armed
    # the real wrapper protocol with ONE PARAMETER PER BUCKET: a backward kernel writes the gradients of a (weight, bias) pair AFTER
    # train_ops._grad_slot(weight, bias) returned; no bucket may be reduced before the kernel that fills it is enqueued (a bucket
    # holding only the first tensor of a pair used to be launched by the second tensor's report - its gradient was reduced without
    # this step's contribution and leaked into the next step)
    from mm_diffusion import train_ops
    for step in range(2):
        if step == 0:
            torch.manual_seed(1)
            params = [torch.nn.Parameter(torch.randn(*s)) for s in [(4, 4), (16,)] * 4]   # 4 (weight, bias) pairs of equal sizes: one tensor per bucket
            opt = FlatAdamW(params, lr=1e-3, grad_buckets=len(params))
            assert len(opt.buckets) == len(params)
        opt.zero_grad()
        g = torch.Generator().manual_seed(200 + 10 * step + rank)
        grads = [torch.randn(*p.shape, generator=g) for p in params]
        expect = torch.cat([x.reshape(-1) for x in grads])
        dist.all_reduce(expect, op=dist.ReduceOp.SUM)
        expect.div_(world)
        opt.arm_overlap()
        for j in reversed(range(0, len(params), 2)):               # backward order, one "kernel" per pair
            slot = train_ops._grad_slot(params[j], params[j + 1])
            assert slot is not None
            slot[0].add_(grads[j])                                 # the kernel: accumulates into .grad after the wrapper's report
            slot[1].add_(grads[j + 1])
        opt.all_reduce_grads()
        res.append((torch.equal(opt.grad, expect), True))
        assert not opt._inflight and not opt._armed
    dist.barrier()
    q.put((rank, [r[0] for r in res], [r[1] for r in res]))
    dist.destroy_process_group()


def test_gloo_world2_bucketed_gradient_allreduce_equals_flat():
    """FlatAdamW gradient buckets (1 / 3 / 7, with and without the overlap protocol driven in backward order): bitwise the result of ONE
    all-reduce of the flat buffer; with overlap armed every bucket but the first-parameter one is in flight before all_reduce_grads()."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, oks, early in res:
        assert oks == [True] * 6              # 4 bucket configurations + 2 steps of the real wrapper protocol, one parameter per bucket
        assert early == [True] * 6            # armed: every bucket was in flight before all_reduce_grads() (flush after the last report)
