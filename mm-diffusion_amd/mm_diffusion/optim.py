"""Flat-buffer AdamW + EMA for the training step (reference: th.optim.AdamW in TrainLoop, multimodal_train_util.py:151-160,
EMA update nn.py:128-138).  All parameters live as views of ONE fp32 buffer, so the optimizer step is one kernel launch
(mmd_adamw_step) and the data-parallel gradient all-reduce is one RCCL call on one flat buffer (the reference's DDP
issues 5 buckets; its sync_params 1046 broadcasts)."""
import torch
import torch.distributed as dist

from . import ops


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_rates=()):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:           # re-home every parameter (and its .grad) into the flat buffers
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.ema_rates = list(ema_rates)
        self.ema_params = [self.flat.clone() for _ in self.ema_rates]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.steps = 0

    def zero_grad(self):
        self.grad.zero_()

    def all_reduce_grads(self):
        """Data-parallel gradient mean over ranks: ONE all-reduce of the flat fp32 gradient buffer (RCCL over xGMI)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
            self.grad.div_(dist.get_world_size())

    def step(self):
        self.steps += 1
        # autograd accumulates into p.grad in place, so self.grad already holds the flat gradient
        for p in self.params:
            if p.grad is not None and p.grad.data_ptr() < self.grad.data_ptr() or p.grad.data_ptr() >= self.grad.data_ptr() + self.grad.numel() * 4:
                raise RuntimeError("a parameter's .grad was replaced; FlatAdamW needs in-place gradient accumulation")
        ema0 = self.ema_params[0] if self.ema_params else None
        ops.adamw_step(self.flat, self.grad, self.m, self.v, ema0, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.weight_decay, self.steps, ema_rate=self.ema_rates[0] if self.ema_rates else 0.0)
        for rate, ema in zip(self.ema_rates[1:], self.ema_params[1:]):
            ema.mul_(rate).add_(self.flat, alpha=1 - rate)
