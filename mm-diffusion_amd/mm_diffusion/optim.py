"""Flat-buffer AdamW + EMA for the training step (reference: th.optim.AdamW in TrainLoop, multimodal_train_util.py:151-160,
EMA update nn.py:128-138).  All parameters live as views of ONE fp32 buffer, so the optimizer step is one kernel launch
(mmd_adamw_step) and the data-parallel gradient all-reduce is one RCCL call on one flat buffer (the reference's DDP
issues 5 buckets; its sync_params 1046 broadcasts)."""
import ctypes

import torch
import torch.distributed as dist

from . import _hip as H
from . import ops


class _PackDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("bwd", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("nt", ctypes.c_int), ("block_start", ctypes.c_int)]


class WeightPacker:
    """Keeps the GEMM-operand copies of every conv weight (forward [Cout, tap*Cin] and data-gradient [Cin, tap*Cout], in the
    activation dtype) current with ONE kernel launch per optimizer step (mmd_pack_conv_weights) and publishes them as
    `param._mmd_packed` for train_ops.ConvFn.  Parameters must stay where they are (views of FlatAdamW's flat buffer)."""

    def __init__(self, params, dtype):
        self.dtype = dtype
        convs = [p for p in params if p.dim() >= 3]
        self.n = len(convs)
        if not self.n:
            return
        dev = convs[0].device
        total = sum(p.numel() for p in convs)
        self.fwd = torch.empty(total, dtype=dtype, device=dev)
        self.bwd = torch.empty(total, dtype=dtype, device=dev)
        descs = (_PackDesc * self.n)()
        off = blocks = 0
        es = self.fwd.element_size()
        for i, p in enumerate(convs):
            Cout, Cin = p.shape[0], p.shape[1]
            nt = p.numel() // (Cout * Cin)
            descs[i] = _PackDesc(p.data_ptr(), self.fwd.data_ptr() + off * es, self.bwd.data_ptr() + off * es, Cout, Cin, nt, blocks)
            p._mmd_packed = (self.fwd[off:off + p.numel()].view(Cout, nt * Cin), self.bwd[off:off + p.numel()].view(Cin, nt * Cout))
            off += p.numel()
            blocks += (p.numel() + 2047) // 2048
        self.blocks = blocks
        raw = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        self.descs = raw.to(dev)
        self.refresh()
        # gradient side: wgrad accumulates in the packed layout (coalesced atomics) and one launch per step folds it into .grad
        self.gpacked = torch.zeros(total, dtype=torch.float32, device=dev)
        gdescs = (_PackDesc * self.n)()
        off = blocks = 0
        for i, p in enumerate(convs):
            Cout, Cin = p.shape[0], p.shape[1]
            nt = p.numel() // (Cout * Cin)
            gdescs[i] = _PackDesc(p.grad.data_ptr(), self.gpacked.data_ptr() + off * 4, None, Cout, Cin, nt, blocks)
            p._mmd_wgrad = self.gpacked[off:off + p.numel()].view(Cout, nt * Cin)
            off += p.numel()
            blocks += (p.numel() + 2047) // 2048
        self.gdescs = torch.frombuffer(bytearray(bytes(gdescs)), dtype=torch.uint8).to(dev)

    def fold_grads(self):
        """.grad += packed wgrad accumulators (then cleared); call once after backward, before the all-reduce / optimizer step."""
        if self.n:
            H.call("mmd_unpack_conv_grads", self.gdescs.data_ptr(), self.n, self.blocks, H.stream_handle())

    def refresh(self):
        if self.n:
            H.call("mmd_pack_conv_weights", H.BF16 if self.dtype == torch.bfloat16 else H.F32, self.descs.data_ptr(), self.n, self.blocks,
                   H.stream_handle())


class FlatAdamW:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_rates=(), pack_dtype=None):
        """pack_dtype: the model's activation dtype (model.dtype) - enables the one-launch conv-weight re-pack after every step."""
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
        self.packer = WeightPacker(self.params, pack_dtype) if pack_dtype is not None and self.params[0].is_cuda else None

    def zero_grad(self):
        self.grad.zero_()

    def fold_grads(self):
        """Fold the packed conv-weight gradient accumulators into .grad (once per step: all_reduce_grads() or step() does it)."""
        if self.packer is not None and not getattr(self, "_folded", False):
            self.packer.fold_grads()
        self._folded = True

    def all_reduce_grads(self):
        """Data-parallel gradient mean over ranks: ONE all-reduce of the flat fp32 gradient buffer (RCCL over xGMI).  Folds the
        packed conv-weight gradient accumulators into .grad first."""
        self.fold_grads()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
            self.grad.div_(dist.get_world_size())

    def step(self):
        self.fold_grads()             # no-op when all_reduce_grads already folded (the accumulators are cleared by the fold)
        self.steps += 1
        # autograd accumulates into p.grad in place, so self.grad already holds the flat gradient
        lo, hi = self.grad.data_ptr(), self.grad.data_ptr() + self.grad.numel() * 4
        for p in self.params:
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError("a parameter's .grad was replaced or dropped; FlatAdamW needs in-place gradient accumulation")
        ema0 = self.ema_params[0] if self.ema_params else None
        ops.adamw_step(self.flat, self.grad, self.m, self.v, ema0, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.weight_decay, self.steps, ema_rate=self.ema_rates[0] if self.ema_rates else 0.0)
        for rate, ema in zip(self.ema_rates[1:], self.ema_params[1:]):
            ema.mul_(rate).add_(self.flat, alpha=1 - rate)
        if self.packer is not None:
            self.packer.refresh()
        self._folded = False
        # the kernel wrote the parameters through the flat buffer: tell autograd / the inference engine (which keys its packed
        # weights on the parameters' version counters, engine.UNetEngine.stale) that every parameter changed
        bump = getattr(torch._C, "_increment_version", None)
        if bump is not None:
            for p in self.params:
                bump(p)
